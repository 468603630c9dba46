"""Shape-compatible, random-init torch modules and synthetic calibration data.
Plumbing only: they give the pruner the reference's parameter names, shapes and
block call signatures (SURVEY.md §8 a-M); the product is the HIP hot path."""

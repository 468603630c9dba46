"""Make one of the build's shape modules look like the reference's own modules to the pruners:
no `stage_plan()` attribute, `forward` unchanged (it still is the composition of the stage
functions, captured before the attribute goes).  A reference user who swaps the import
 hands the pruners exactly such a model — LAVIS's Blip2T5
(LAVIS/lavis/models/blip2_models/blip2_t5.py:116-168) has block lists and a forward, nothing
else — so this is what the tests and `bench.py --unstaged` drive the un-staged path
(pruners/hooked_prefix.py) with."""
import contextlib


@contextlib.contextmanager
def hidden_stage_plan(*classes):
    saved = []
    for cls in classes:
        plan = cls.__dict__.get("stage_plan")
        if plan is None:
            continue

        def composed(self, samples, _plan=plan):
            held = self.__dict__.get("_unstaged_fns")
            if held is None or held[0] != id(self):   # (the stage functions close over `self`: built
                held = self.__dict__["_unstaged_fns"] = (id(self), [fn for _, _, fn in _plan(self)])
            fns = held[1]                             #  once per model; a deep copy builds its own)
            state = samples
            for fn in fns:
                state = fn(state)
            return state

        # the entry points that are the composition of the stages (`forward`; EVACLIP: `predict`):
        # replaced by the class's reference-style forward where it has one (the same ops written
        # as the reference writes its forward), by the composition otherwise
        entries = {name: cls.__dict__[name] for name in ("forward", "predict") if name in cls.__dict__}
        saved.append((cls, plan, entries))
        for name in entries:
            ref = cls.__dict__.get("reference_" + name)
            setattr(cls, name, ref if ref is not None else composed)
        del cls.stage_plan
    try:
        yield
    finally:
        for cls, plan, entries in saved:
            cls.stage_plan = plan
            for name, fn in entries.items():
                setattr(cls, name, fn)


_KEEP = []


def hide_stage_plan(*classes):
    """The same, for the rest of the process (child processes of the tests, bench.py --unstaged)."""
    ctx = hidden_stage_plan(*classes)
    ctx.__enter__()
    _KEEP.append(ctx)         # (a collected generator would run its `finally` and undo the hiding)
    return ctx

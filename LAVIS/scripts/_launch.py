"""Launcher table: every script under LAVIS/scripts/** is a stub that names one entry here.
Same role as the reference's launch scripts (a `subprocess.call` of `python -m
torch.distributed.run ... evaluate_*.py ...` with the method / score / ratio strings and job id of
that experiment), pointed at the build's harness (ecoflap_amd/harness.py).  Usage of every
script: `python <script> GPU PORT [extra harness flags]` (the reference's two positional
arguments)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# blocks per tower of the build's shapes (the harness spec strings are "<blocks>-<keep>-1.0-1.0")
BLOCKS = {"t5": 24, "vit_g": 39, "vit_b": 12}

# name -> (shape, pruner, keep ratio, flags, job-id template, reference lines)
#   flags: score / granularity / cap (= 1 - keep + 0.1) / bs / first_stage / is_global /
#          per_model / iteration / eps sweep / table (a stage-1 table is taken from argv[3] if given)
JOBS = {
    "blip2/ecoflap_zeroth": ("blip2", "blipt5_wanda_pruner", 0.5,
                             dict(score="MEZO-GradOnly_sum", granularity="block", cap=True, bs=8),
                             "cc3m-{pruner}_{spec}_{score}{cap}_{granularity}_bs{bs}"),
    "blip2/ecoflap_first": ("blip2", "blipt5_wanda_pruner", 0.5,
                            dict(score="GradMagAbs_sum", granularity="block", cap=True, first_stage=128),
                            "cc3m-{pruner}_{spec}_{score}{cap}_{granularity}"),
    "blip2/ecoflap_zeroth_eps": ("blip2", "blipt5_wanda_pruner", 0.5,
                                 dict(score="MEZO-GradOnly_sum", granularity="block", cap=True,
                                      eps_sweep=[1e-1, 1e-2, 1e-4]),
                                 "cc3m-{pruner}_{spec}_{score}{cap}_{granularity}_eps{eps}"),
    "blip2/ecoflap_sparsegpt_zeroth": ("blip2", "blipt5_sparsegpt_pruner", 0.4,
                                       dict(score="MEZO-GradOnly_sum", granularity="block", cap_value=0.7,
                                            bs=1, table=True),
                                       "cc3m-{pruner}_{spec}_olmezo-gradient_sum0.7_block"),
    "blip2/ecoflap_sparsegpt_first": ("blip2", "blipt5_sparsegpt_pruner", 0.4,
                                      dict(score="GradMagAbs_sum", granularity="block", cap_value=0.7,
                                           first_stage=128, table=True),
                                      "cc3m-{pruner}_{spec}_aobd_sum0.7_block"),
    "blip2/wanda": ("blip2", "blipt5_wanda_pruner", 0.5, {}, "cc3m-{pruner}_{spec}"),
    "blip2/sparsegpt": ("blip2", "blipt5_sparsegpt_pruner", 0.5, {}, "cc3m-{pruner}_{spec}"),
    "blip2/mag": ("blip2", "blipt5_global_mag_pruner", 0.5, dict(is_global=True),
                  "cc3m-{pruner}_{spec}_global"),
    "blip2/iterative_global_gradient": ("blip2", "blipt5_global_gradmagabs_pruner", 0.5,
                                        dict(is_global=True, per_model=True, iteration=3),
                                        "cc3m-{pruner}_{spec}_iteration{iteration}_global_per_model"),
    "t5/ecoflap": ("t5", "t5_wanda_pruner", 0.5,
                   dict(score="MEZO-GradOnly_avg", granularity="block", cap=True, bs=8),
                   "cc3m-{pruner}_{spec}_{score}{cap}_{granularity}_bs{bs}"),
    "t5/wanda": ("t5", "t5_wanda_pruner", 0.5, {}, "cc3m-{pruner}_{spec}"),
    "eva_clip/ecoflap": ("vit", "vit_wanda_pruner", 0.5,
                         dict(score="MEZO-GradOnly_sum", granularity="block", cap=True, bs=8),
                         "imgn-{pruner}_{spec}_{score}{cap}_{granularity}_bs{bs}"),
    "eva_clip/wanda": ("vit", "vit_wanda_pruner", 0.5, {}, "imgn-{pruner}_{spec}"),
}


def harness_flags(shape, pruner, keep, opt, job_template, eps=None, table=None):
    spec = f"{keep}-1.0-1.0"
    cap = opt.get("cap_value", round(1.0 - keep + 0.1, 1) if opt.get("cap") else None)
    fields = dict(pruner=pruner, spec=spec, score=opt.get("score"), cap=cap,
                  granularity=opt.get("granularity"), bs=opt.get("bs"), eps=eps,
                  iteration=opt.get("iteration"))
    flags = [f"--pruning_method '{pruner}'", "--save_pruned_model"]
    if table is not None:
        flags.append(f"--sparsity_dict {table}")
    elif opt.get("score"):
        flags += [f"--score_method {opt['score']}", f"--sparsity_ratio_granularity {opt['granularity']}",
                  f"--max_sparsity_per_layer {cap}"]
    if opt.get("bs"):
        flags.append(f"--prunining_dataset_batch_size {opt['bs']}")
    if opt.get("first_stage") and table is None:
        flags.append(f"--num_data_first_stage {opt['first_stage']}")
    if eps is not None:
        flags.append(f"--noise_eps {eps}")
    if opt.get("is_global"):
        flags.append("--is_global")
    if opt.get("per_model"):
        flags.append("--prune_per_model")
    if opt.get("iteration"):
        flags.append(f"--iteration {opt['iteration']}")
    if shape in ("blip2", "t5"):
        flags.append(f"--t5_prune_spec {BLOCKS['t5']}-{spec}")
    if shape == "blip2":
        flags.append(f"--vit_prune_spec {BLOCKS['vit_g']}-{spec}")
    if shape == "vit":
        flags.append(f"--vit_prune_spec {BLOCKS['vit_b']}-{spec}")
    flags.append(f"--job_id '{job_template.format(**fields)}'")
    return " ".join(flags)


def launch(shape, flags):
    gpu = sys.argv[1] if len(sys.argv) > 1 else "0"
    port = sys.argv[2] if len(sys.argv) > 2 else "12341"
    extra = " ".join(sys.argv[3:])
    program = (f"HIP_VISIBLE_DEVICES={gpu} TENSILE_STREAMK_DATA_PARALLEL=1 {sys.executable} -m torch.distributed.run"
               f" --nproc_per_node=1 --master-addr 127.0.0.1 --master_port {port}"
               f" -m ecoflap_amd.harness --shape {shape} {flags} {extra}")
    print(program)
    return subprocess.call(program, shell=True, cwd=ROOT)


def run(name):
    shape, pruner, keep, opt, template = JOBS[name]
    table = None
    if opt.get("table") and len(sys.argv) > 3 and not sys.argv[3].startswith("-"):
        table = sys.argv.pop(3)       # a stage-1 table written by an earlier ecoflap_* run
    rc = 0
    for eps in opt.get("eps_sweep", [None]):
        rc |= launch(shape, harness_flags(shape, pruner, keep, opt, template, eps=eps, table=table))
    return rc

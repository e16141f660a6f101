#!/usr/bin/env python
"""Copy a round profile from gpurun_out/<tag>/ (tools/profile_round.sh + tools/pmc_run.sh gpurun_out/<tag>_pmc ...) into profiles/<tag>_*,
write profiles/<tag>_pmc_counters.txt with the derived ratios, drop the files of the tag it replaces and retarget the documents:
    python tools/install_profile.py <tag> [<old tag>]"""
import glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, ROOT)
tag, old = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
FILES = ("bench_train.json", "bench_forward.json", "bench_fp16.json", "kernel_stats.csv", "forward_kernel_stats.csv", "profiled_line.json",
         "profiled_forward_line.json", "pmc_FETCH_SIZE_per_kernel_mean_kb.csv", "pmc_WRITE_SIZE_per_kernel_mean_kb.csv", "pmc_traffic.json",
         "glu_layer_durations.csv", "bench_c3.json", "bench_c5.json", "pytest_gpu.log")
NAMES = {"glu_fwd_static_kernel<": "glu_fwd_static_kernel (inference launch of the fused layer)",
         "glu_fwd_static_z_kernel<": "glu_fwd_static_z_kernel (training launch, z saved)",
         "gemm_tm_kernelIDF16bLi8ELi1E": "gemm_tm_kernel<bf16, NT=8, MODE=1> (residual backward, per layer)",
         "gemm_tm_kernelIDF16bLi6ELi2E": "gemm_tm_kernel<bf16, NT=6, MODE=2> (gate backward, per layer)",
         "gemm_tm_kernelIDF16bLi8ELi3E": "gemm_tm_kernel<bf16, NT=8, MODE=3> (the head's skip contraction, K = 4608)",
         "gemm_tm_kernelIDF16bLi2ELi0E": "gemm_tm_kernel<bf16, NT=2, MODE=0> (dc, K = 9216)",
         "gemm_tn_static_kernel": "gemm_tn_static_kernel (every weight gradient of the step)",
         "glu_bwd_pair_kernel": "glu_bwd_pair_kernel (residual(l) + gate(l-1) in one launch: the 16-bit backward sweep)",
         "head_fwd_kernel": "head_fwd_kernel<..., FROM_H0> (GEMM 1, GEMM 2, fused CE)", "head_bwd_kernel": "head_bwd_kernel"}
if old:
    for fn in glob.glob(f"profiles/{old}_*"):
        os.remove(fn)
for fn in FILES:
    if os.path.exists(f"gpurun_out/{tag}/{fn}"):
        shutil.copy(f"gpurun_out/{tag}/{fn}", f"profiles/{tag}_{fn}")
blocks, cur = {}, None
for line in open(f"gpurun_out/{tag}_pmc/summary.txt"):
    m = re.match(r"== kernels matching '(.*)'", line.strip())
    if m:
        cur = m.group(1); blocks[cur] = []
    elif cur and line.strip():
        blocks[cur].append(line.rstrip())
out = ["# rocprofv3 --kernel-trace --pmc <set> (six separate passes, tools/pmc_run.sh) over `python3 bench.py --steps 2 --warmup 1 --no-cpu --no-ar`,",
       f"# final tree of the round (csrc hash in {tag}_pmc_traffic.json), one MI355X.  Kernel selection by substring of the profiler's kernel name.",
       "# derived: MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)  (SQ_BUSY_CYCLES counts per shader array: 32 arrays x 32 SIMDs);",
       "#          L2 hit = TCC_HIT / TCC_REQ;  LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024", ""]
for k, lines in blocks.items():
    if not lines:          # (a kernel that does not run in this configuration any more, e.g. the stand-alone residual launch)
        continue
    v = {l.split()[0]: float(l.split()[2]) for l in lines}
    out.append("== " + NAMES.get(k, k))
    out.append("   derived: MFMA busy %.1f %%, L2 hit rate %.0f %%, LDS bank-conflict share %.1f %%, waves waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) %.0f %%, "
               "HBM traffic %.1f MB per launch" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / (32 * v["SQ_BUSY_CYCLES"]) * 100, v["TCC_HIT"] / v["TCC_REQ"] * 100,
                                                    v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1) * 100,
                                                    v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"] * 100, (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / 1e6))
    out += ["   " + l for l in lines] + [""]
open(f"profiles/{tag}_pmc_counters.txt", "w").write("\n".join(out))
if old:
    for p in ("DESIGN.md", "README.md", "profiles/README.md", "profiles/EXPERIMENT_LOG.md"):
        s = open(p).read()
        s = s.replace(old + "_", tag + "_").replace("`" + old, "`" + tag).replace("profile_round.sh " + old, "profile_round.sh " + tag)
        open(p, "w").write(s)
import bench
d = json.loads(open(f"gpurun_out/{tag}/bench_train.json").read().strip().splitlines()[-1])
fi = d["forward_inference"]
print("csrc hash", bench.csrc_hash(), "profile", json.load(open(f"profiles/{tag}_pmc_traffic.json"))["csrc_hash"])
ms = lambda k: d[k]["avg_launch_ms"] * 1e3 if k in d else float("nan")      # (a family that did not run in this build: nan)
print("train %.3f ms %.2f M/s | forward %.3f ms, layer %.1f us frac %.3f, whole %.3f | wgrad %.3f ms | glu_z %.1f us | gate %.1f res %.1f pair %.1f us | AR %.2f kHz | cpu %.0f" % (
    d["ms_per_step"], d["value"] / 1e6, fi["ms_per_step"], fi["roofline"]["avg_launch_ms"] * 1e3, fi["roofline"]["frac"], fi["roofline_whole"]["frac"],
    d["roofline_wgrad"]["avg_launch_ms"], ms("roofline_glu_fwd_z"), ms("roofline_gate_bwd"), ms("roofline_residual_bwd"), ms("roofline_bwd_pair"),
    d["autoregressive"]["value"], d["cpu_baseline"]["value"]))
for fn in ("bench_forward.json", "bench_fp16.json"):
    print(fn, json.loads(open(f"gpurun_out/{tag}/{fn}").read().strip().splitlines()[-1])["ms_per_step"])
for fn in ("bench_c3.json", "bench_c5.json"):
    r = json.loads(open(f"gpurun_out/{tag}/{fn}").read().strip().splitlines()[-1])
    print(fn, r["dtype"], "train", r["ms_per_step"], "forward", r["forward_inference"]["ms_per_step"], "step roofline", r["roofline_step"]["hbm_frac"], r["roofline_step"]["mfma_frac"])
print("\n".join(l for l in out if "derived:" in l))

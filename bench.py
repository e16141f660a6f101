#!/usr/bin/env python
"""bench.py -- teacher-forced throughput of the decoder hot path on synthetic 16 kHz clips.

Workload (BASELINE.json configs[1], "C2"): IN-WAE decoder dimensions (hps/inae_hp.json: R=256, G=368, S=256,
Cc=64, Cg=64, k=3, upsample x320) with 24 layers / 2 stacks, batch 8 x 8000 samples per GPU, bf16 storage with
fp32 accumulate.  A step = one teacher-forced pass over one batch: weight-norm + fragment packing of all
parameters, conditioning upsample, speaker projection, first-conv gather, 24 fused gated layers, head and the
fused shifted cross-entropy (mean loss on device), then the full backward pass (head, 24 x {du/dz, weight gradients,
dx}, conditioning/upsample gradients, weight-norm backward), [N > 1: bucketed RCCL all-reduce of the gradient arena]
and the fused clip_grad_norm_ + Adam + EMA update.  Inputs are resident in HBM before the timed region.
`--mode forward` times the teacher-forced forward pass alone.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c5] [--dtype bf16|fp16|fp32] [--mode train|forward] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Multi-GPU: one process per GPU, each rank trains on its own shard of the global
batch (weak scaling: 8 clips per GPU), gradients all-reduced over RCCL (started inside backward, on a side stream),
barrier + synchronize on both sides, MAX over ranks.  `python bench.py --gpus N` without a launcher starts the N ranks
itself (torch.distributed.run as a child process, before this process has touched a GPU) and relays their line.
"""
import subprocess
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C2 = dict(layers=24, stacks=2, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=153,
          upsample_scales=[4, 4, 4, 5], cin_pad=0)
B_PER_GPU, T = 8, 8000
# --config: the three trainable configurations BASELINE.json names.  c2 (the default, configs[1]) is the one the metric is quoted on;
# c3 = configs[2], hps/vqwae.json in full (encoder + VQ + 20-layer decoder), the per-GPU shard of its global batch 64 on 8 GPUs;
# c5 = configs[4], 48 layers x 512 channels in fp16, the per-GPU shard of batch 128.
CONFIGS = {
    "c2": dict(cfg=C2, B=8, T=8000, dtype="bf16", encoder=False, salt=5,
               name="C2: IN-WAE decoder dims (R256 G368 S256 Cc64 Cg64 k3), 24 layers/2 stacks"),
    "c3": dict(cfg=dict(layers=20, stacks=2, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=153, upsample_scales=[4, 4, 8, 5],
                        encoder_hid=256, c_in=39, K=256, cin_pad=0), B=8, T=5120, dtype="bf16", encoder=True, salt=7,
               name="C3: hps/vqwae.json in full (encoder 39->256 x10 blocks, VQ K=256, upsample x640, 20-layer decoder R=G=S=256)"),
    "c5": dict(cfg=dict(layers=48, stacks=4, R=512, G=512, S=512, O=256, Cc=64, Cg=32, k=3, n_speakers=8, upsample_scales=[4, 4, 4, 5],
                        cin_pad=0), B=16, T=5120, dtype="fp16", encoder=False, salt=3,
               name="C5: 48 layers/4 stacks, R=G=S=512 decoder (Cc64 Cg32 k3)"),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TF = 2500.0      # dense bf16
FP32_MFMA_PEAK_TF = 157.3


def synth_inputs(rank, device, conf=None, on_device=True):
    """(class ids (B,T), conditioning, speaker ids): the conditioning is the latent sequence (B, Cc, T/hop) for a decoder-only
    configuration and the encoder's input features (B, c_in, T/160: hps/vqwae.json's 100 frames/s at 16 kHz) for c3."""
    import numpy as np
    import torch
    conf = conf or CONFIGS["c2"]
    cfg, B, T_ = conf["cfg"], conf["B"], conf["T"]
    hop = int(np.prod(cfg["upsample_scales"]))
    rng = np.random.default_rng(1234 + rank)
    x = torch.from_numpy(rng.integers(0, cfg["O"], size=(B, T_), dtype=np.int64))
    if conf["encoder"]:
        lat = torch.from_numpy(rng.standard_normal((B, cfg["c_in"], T_ // (hop // 4))).astype(np.float32))   # the encoder strides by 4 (vqvae_model.py:32-40)
    else:
        lat = torch.from_numpy(rng.standard_normal((B, cfg["Cc"], T_ // hop)).astype(np.float32))
    g = torch.from_numpy(rng.integers(0, cfg["n_speakers"], size=(B,), dtype=np.int64))
    if not on_device:
        return x, lat, g
    return x.to(device), lat.to(device), g.to(device)


AR_CFG = dict(layers=20, stacks=2, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=153,
              upsample_scales=[4, 4, 8, 5], cin_pad=0)


def ar_leg(device, T_ar=4000, cpu=True, full_clip=True):
    """Second half of BASELINE.json's metric: autoregressive kHz of synthesis.py's incremental_forward on one GPU, one
    utterance (config C4: hps/vqwae.json decoder, 16 kHz; a 0.25 s prefix of the 10 s clip -- the per-sample cost is
    constant), categorical sampling as in the reference (wavenet.py:300-338).  Two untimed runs, then one timed run.
    Roofline: AR is latency-bound; the bound quoted is streaming every effective weight once per sample
    (5.9 M x e bytes at 8 TB/s).  cpu_baseline: the oracle's incremental loop on a 1600-sample prefix."""
    import torch
    from oracle import wae_oracle as O          # closed-form weights + the cpu_baseline leg only
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg = AR_CFG
    out = {}
    sd = O.make_state_dict(dict(cfg), salt=7, with_encoder=False)
    n_w = sum(v.numel() for k, v in sd.items() if k.endswith("weight_v") and ("conv_layers" in k or "last_conv" in k))
    for dt in ("fp32", "bf16"):
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dt, device=str(device))
        eng.load_state_dict(sd)
        gen = torch.Generator(device="cpu").manual_seed(1234)
        lat = torch.randn(1, 64, T_ar // 640 + 1, generator=gen)[:, :, :max(T_ar // 640, 1)].to(device)
        Tg = lat.shape[-1] * 640
        gid = torch.zeros(1, dtype=torch.int64, device=device)
        for _ in range(2):      # (two untimed runs: the first call of an engine packs the decode weights and sizes its buffers)
            eng.incremental_forward(lat, gid, Tg, mode="sample")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.incremental_forward(lat, gid, Tg, mode="sample")
        torch.cuda.synchronize()
        out[dt] = Tg / (time.perf_counter() - t0) / 1e3
    best = "bf16" if out["bf16"] >= out["fp32"] else "fp32"
    full = None
    if full_clip:
        # BASELINE.json configs[3] as named: ONE 16 kHz 10 s clip = 250 latent frames -> 160 000 samples, end to end, best dtype
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=best, device=str(device))
        eng.load_state_dict(sd)
        gen = torch.Generator(device="cpu").manual_seed(4321)
        lat = torch.randn(1, 64, 250, generator=gen).to(device)
        gid = torch.zeros(1, dtype=torch.int64, device=device)
        eng.incremental_forward(lat[:, :, :2].contiguous(), gid, 1280, mode="sample")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = eng.incremental_forward(lat, gid, 160000, mode="sample")
        torch.cuda.synchronize()
        dtf = time.perf_counter() - t0
        idx = o["idx"][0]
        full = {"samples": 160000, "seconds": dtf, "khz": 160000 / dtf / 1e3, "dtype": best, "realtime_factor": 160000 / dtf / 16000.0,
                "distinct_classes": int(torch.unique(idx).numel()),
                "note": "the whole 10 s clip in one persistent launch (250 latent frames x 640); the headline value above is the "
                        "short prefix run"}
    us = 1e3 / out[best]
    wbytes = n_w * (2 if best == "bf16" else 4)
    stream_us = wbytes / (HBM_PEAK_GBS * 1e9) * 1e6
    # The floor of this kernel is the chain of cross-CU hand-overs, not bytes: a sample is 20 strictly sequential layers, each of which
    # exchanges the members' shares of x' twice (reduce-scatter + all-gather over {sequence number, fp32} granules inside one XCD).
    # Measured inside the kernel with every request removed (timing-only ablation, profiles/r03_ar_stamps.txt, EXPERIMENT_LOG "Round 3:
    # ar_coop"): the two hand-overs of a layer take 1.76 k clocks at the 2.09 GHz this kernel holds (68 k clocks per 32.6-us sample);
    # tools/handover_probe.hip prices one store -> L2 -> poll hand-over at ~0.9 k clocks on an idle XCD.
    HANDOVER_CLOCKS_PER_LAYER, AR_CLOCK_GHZ = 1760.0, 2.09
    floor_us = cfg["layers"] * HANDOVER_CLOCKS_PER_LAYER / (AR_CLOCK_GHZ * 1e3)
    res = {"metric": "autoregressive kHz (synthesis.py incremental_forward, 1 utterance, 1 GPU)", "value": out[best],
           "unit": "kHz", "fp32_khz": out["fp32"], "bf16_khz": out["bf16"], "samples": Tg, "us_per_sample": us,
           "realtime_factor": out[best] / 16.0,
           "roofline": {"bound": "latency (the chain of cross-CU hand-overs: 2 per layer, %d layers per sample)" % cfg["layers"],
                        "achieved": us, "peak": floor_us, "unit": "us/sample", "frac": floor_us / us,
                        "handover_clocks_per_layer": HANDOVER_CLOCKS_PER_LAYER, "kernel_clock_ghz": AR_CLOCK_GHZ,
                        "note": "exchange-latency floor = layers x 1.76 k clocks of hand-over at 2.09 GHz (measured in-kernel with every "
                                "request ablated); the rest of a sample is the per-layer GEMV + gate + barriers between the hand-overs",
                        "weight_streaming_us": stream_us, "weight_bytes_per_sample": wbytes,
                        "weight_streaming_note": "SURVEY 8(d)'s stated bound -- every effective weight once per sample (%d x %d B) / 8 TB/s "
                                                 "-- no longer describes the 16-bit kernel: every layer's weights stay on chip for the whole clip "
                                                 "(LDS + registers), nothing is streamed per sample" % (n_w, 2 if best == "bf16" else 4)},
           "config": "C4: hps/vqwae.json decoder (20 layers, R=G=S=256), categorical sampling, one persistent launch"}
    if full is not None:
        res["full_clip"] = full
    # "Replicas only" (the layer chain of one sample is strictly sequential, utterances are independent): aggregate kHz of a
    # batch of utterances on ONE GPU -- 8 utterances on the cooperative kernel (one XCD each), 256 on the one-CU kernel (one CU each)
    batched = {}
    for nb, Tb in ((8, 2560), (256, 640)):      # (8 x 1280 was a 40-ms window: host-side set-up of the call showed in it)
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=best, device=str(device))
        eng.load_state_dict(sd)
        gen = torch.Generator(device="cpu").manual_seed(99 + nb)
        lat = torch.randn(nb, 64, Tb // 640, generator=gen).to(device)
        gid = torch.arange(nb, dtype=torch.int64, device=device) % cfg["n_speakers"]
        for _ in range(2):
            eng.incremental_forward(lat, gid, Tb, mode="sample")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.incremental_forward(lat, gid, Tb, mode="sample")
        torch.cuda.synchronize()
        dtb = time.perf_counter() - t0
        batched["%d_utterances" % nb] = {"aggregate_khz": nb * Tb / dtb / 1e3, "khz_per_utterance": Tb / dtb / 1e3, "samples_each": Tb,
                                         "kernel": "ar_coop_fast_kernel (32 CUs of one XCD per utterance)" if nb <= 8 else
                                                   "ar_kernel (one CU per utterance)", "dtype": best}
    res["batched"] = batched
    if cpu:
        nthreads = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(nthreads)
        Tc = 1600
        ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=0)
        g = torch.Generator().manual_seed(1234)
        latc = torch.randn(1, 64, 3, generator=g)
        c_up = O.upsample_forward(sd, latc, cfg["upsample_scales"])[:, :, :Tc].contiguous()
        u = torch.rand(1, Tc, generator=g)
        with torch.no_grad():
            O.incremental_forward(sd, ocfg, c_up[:, :, :32].contiguous(), torch.zeros(1, dtype=torch.long), 32, mode="sample", uniforms=u)
            t0 = time.perf_counter()
            O.incremental_forward(sd, ocfg, c_up, torch.zeros(1, dtype=torch.long), Tc, mode="sample", uniforms=u)
            dtc = time.perf_counter() - t0
        res["cpu_baseline"] = dict(value=Tc / dtc / 1e3, unit="kHz", cores=nthreads, kind="port",
                                   sample=f"oracle/wae_oracle.py incremental_forward (O(1) ring lookup restatement of conv.py:17-62), "
                                          f"{Tc}-sample prefix of the same decoder, torch CPU fp32, {nthreads} threads ({dtc:.1f} s, "
                                          "after a 32-sample warm-up)")
    return res


def cpu_baseline_train(sd, nclips=8, conf=None):
    """Oracle train step (autograd through the CPU restatement + its Adam/EMA), bounded sample of the same workload."""
    import torch
    from oracle import wae_oracle as O
    conf = conf or CONFIGS["c2"]
    cfg, T = conf["cfg"], conf["T"]
    B_PER_GPU = conf["B"]
    nthreads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(nthreads)
    x, lat, g = (a[:nclips] for a in synth_inputs(0, None, conf, on_device=False))
    xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
    ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=0)
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
    sh = {k: v.clone() for k, v in sd.items()}

    def loss_of(xin_, x_, lat_, g_):
        if conf["encoder"]:       # VQVAE.forward (vqvae_model.py:66-72): CE + vq_loss
            y_, vq_, _, _ = O.vqvae_forward(psd, ocfg, xin_, lat_, g_)
            return O.masked_ce_loss(y_, x_.unsqueeze(-1), torch.full((x_.shape[0],), T)) + vq_
        return O.masked_ce_loss(O.wavenet_forward(psd, ocfg, xin_, lat_, g_), x_.unsqueeze(-1), torch.full((x_.shape[0],), T))
    # warm-up on one clip (thread pool, oneDNN primitive caches), then ONE timed pass over the sample
    loss_of(xin[:1], x[:1], lat[:1], g[:1]).backward()
    for p in psd.values():
        p.grad = None
    t0 = time.perf_counter()
    loss = loss_of(xin, x, lat, g)
    loss.backward()
    with torch.no_grad():
        grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in psd.items()}
        O.clip_adam_ema_step({k: p.data for k, p in psd.items()}, grads, m, v2, sh, 1, 4e-4)
    dt = time.perf_counter() - t0
    return dict(value=nclips * T / dt, unit="samples/s", cores=nthreads, kind="port",
                sample=f"{nclips} of the {B_PER_GPU} clips x {T} samples, 1 train step (forward+CE+autograd backward+clip/Adam/EMA), "
                       f"oracle/wae_oracle.py on torch CPU fp32, {nthreads} threads of {os.cpu_count()} cores ({dt:.1f} s, after a one-clip warm-up)"), float(loss)


def cpu_baseline(sd, nclips=8, conf=None):
    """Oracle (CPU restatement, kind 'port') timed on a bounded sample of the same workload."""
    import torch
    from oracle import wae_oracle as O
    conf = conf or CONFIGS["c2"]
    cfg, T = conf["cfg"], conf["T"]
    B_PER_GPU = conf["B"]
    nthreads = min(os.cpu_count() or 1, 16)       # more threads than this only slows torch's CPU conv1d down
    torch.set_num_threads(nthreads)
    x, lat, g = (a[:nclips] for a in synth_inputs(0, None, conf, on_device=False))
    xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
    ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=0)
    fwd = (lambda a, b_, c_: O.vqvae_forward(sd, ocfg, a, b_, c_)[0]) if conf["encoder"] else (lambda a, b_, c_: O.wavenet_forward(sd, ocfg, a, b_, c_))
    with torch.no_grad():
        fwd(xin[:1], lat[:1], g[:1])       # warm-up (thread pool, oneDNN primitive caches)
        t0 = time.perf_counter()
        y = fwd(xin, lat, g)
        loss = O.masked_ce_loss(y, x.unsqueeze(-1), torch.full((nclips,), T))
        dt = time.perf_counter() - t0
    return dict(value=nclips * T / dt, unit="samples/s", cores=nthreads, kind="port",
                sample=f"{nclips} of the {B_PER_GPU} clips x {T} samples, 1 forward+CE pass, oracle/wae_oracle.py on "
                       f"torch CPU fp32, {nthreads} threads of {os.cpu_count()} cores ({dt:.1f} s)"), float(loss)


def fp32_leg(conf, sd, device, steps=4):
    """The parity-grade mode's throughput: fp32 storage, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), libm gate -- the mode in which the
    north star's 1e-3 tolerance is met at ~1e-6 (tests/test_gpu_parity.py).  Same workload, a few steps, train and forward."""
    import torch
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    eng = WaeEngine(Geometry.from_cfg(conf["cfg"]), dtype="fp32", device=str(device))
    eng.load_state_dict(sd)
    eng.init_optimizer()
    x, lat, g = synth_inputs(0, device, conf)
    xi = x.to(torch.int32)
    fwd = eng.forward if conf["encoder"] else eng.decoder_forward
    out = {}
    for name, fn in (("train", lambda: eng.train_step(xi, lat, g, lengths=None)["loss"]),
                     ("forward", lambda: fwd(xi, lat, g, targets=xi, want_logits=False)["loss"])):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            loss = fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        out[name] = {"ms_per_step": ms, "samples_per_s": conf["B"] * conf["T"] / (ms * 1e-3), "loss": float(loss)}
    fl = conf["cfg"]
    H = fl["G"] // 2
    fwd_flops = (fl["layers"] * 2 * (fl["G"] * fl["R"] * fl["k"] + fl["G"] * fl["Cc"] + H * fl["R"] + H * fl["S"])
                 + 2 * (fl["S"] * fl["S"] + fl["S"] * fl["O"])) * conf["B"] * conf["T"]
    out["train"]["mfma_frac"] = 3 * fwd_flops / (out["train"]["ms_per_step"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF
    out["forward"]["mfma_frac"] = fwd_flops / (out["forward"]["ms_per_step"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF
    out["note"] = ("dtype fp32: fp32 activations / weights / accumulation (the parity gate: logits and latents <= 1e-3 relative, measured "
                   "~1e-6); mfma_frac against the dense fp32 matrix peak %.1f TFLOP/s" % FP32_MFMA_PEAK_TF)
    del eng
    torch.cuda.empty_cache()
    return out


def sub_config_line(name, steps=5, warmup=2, cpu=True):
    """`bench.py --config <name>` as a child process, its line cut down to what the default line embeds."""
    cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", str(steps), "--warmup", str(warmup), "--no-ar",
           "--no-fp32", "--no-sub"] + ([] if cpu else ["--no-cpu"])
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            return {"error": "rc %d: %s" % (p.returncode, p.stderr[-400:])}
        d = json.loads(line[-1])
    except (subprocess.TimeoutExpired, ValueError) as e:
        return {"error": repr(e)}
    keep = {k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "loss", "roofline_step", "cpu_baseline")
            if k in d}
    keep["workload"] = d["config"]["workload"]
    r = d["roofline"]
    keep["roofline"] = {k: r[k] for k in ("bound", "kernel", "family", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms",
                                          "ms_per_step", "algorithmic_bytes_per_launch", "mfma_frac", "family_ms_per_step", "clips_per_launch",
                                          "concurrent_launches", "both_chains") if k in r}
    fi = d.get("forward_inference")
    if fi:
        keep["forward_inference"] = {"ms_per_step": fi["ms_per_step"], "value": fi["value"], "roofline_whole_frac": fi["roofline_whole"]["frac"],
                                     "layer_launch_frac": fi["roofline"]["frac"], "layer_launch_ms": fi["roofline"]["avg_launch_ms"]}
    keep["wall_s"] = time.perf_counter() - t0
    return keep


def csrc_hash():
    """sha256 over the kernel sources: profiles/*_pmc_traffic.json carries the hash of the build it was measured on"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "wavenet_autoencoders_amd", "csrc", "*.h*")) + [os.path.join(ROOT, "include", "wae.h")]):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def glu_kernel_name(dtype, save_z):
    """Name of the fused layer kernel a C2 launch runs: the 16-bit modes take the static-schedule instantiation
    (csrc/glu_fwd_static.hip), whose training launches (z saved) and inference launches are differently NAMED entry points."""
    if dtype == "fp32":
        return "glu_fwd_kernel"
    return "glu_fwd_static_z_kernel" if save_z else "glu_fwd_static_kernel"


def kernel_matches(name, profiler_name):
    """Does a rocprofv3 kernel name belong to the kernel `name`?  (name followed by '<' or '(': glu_fwd_static_kernel must not
    match glu_fwd_static_z_kernel; mangled names -- _Z21gemm_tn_stream_kernelI... -- match by substring.)  The last layer's
    launch (WAE_GLU_NO_OUT: x' is dead) is a different instantiation of the same name with about 20 % less traffic; the mean over
    a name's launches is what tools/profile_round.sh stores."""
    if name.startswith("gemm_tm_kernel:"):     # "gemm_tm_kernel:<MODE>": the per-layer backward launches are instantiations of ONE kernel
        import re                               # template, told apart by its MODE argument (mangled: gemm_tm_kernelI<E>Li<NT>ELi<MODE>E...)
        m = re.search(r"gemm_tm_kernelI\w+?Li(\d+)ELi(\d+)E", profiler_name) or re.search(r"gemm_tm_kernel<[^,]+, *(\d+), *(\d+)", profiler_name)
        return bool(m) and m.group(2) == name.split(":")[1]
    for sep in ("<", "("):
        if name + sep in profiler_name:
            return True
    # tools/profile_round.sh aggregates the static layer kernel's launches under the bare entry-point name ("void glu_fwd_static_z_kernel")
    if profiler_name.split("<")[0].split("(")[0].split()[-1:] == [name]:
        return True
    return profiler_name.startswith("_Z") and name + "I" in profiler_name


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (this process has not
    initialised the GPU and never will), relay its output, exit with its code."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS),
                    help="c2 (default): the configuration the metric is quoted on; c3: hps/vqwae.json in full, 8 x 5120 per GPU; "
                         "c5: 48 layers x 512 channels, 16 x 5120 per GPU, fp16")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16", "fp32"], help="default: the configuration's (c2, c3: bf16; c5: fp16)")
    ap.add_argument("--mode", default="train", choices=["train", "forward"])
    ap.add_argument("--ar-short", action="store_true", help="autoregressive leg: skip the full 160 000-sample C4 clip (about 24 s)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-ar", action="store_true", help="skip the autoregressive leg (BASELINE config C4, rank 0 only)")
    ap.add_argument("--no-fp32", action="store_true", help="skip the short fp32 (parity-grade mode) leg of the default line")
    ap.add_argument("--no-sub", action="store_true", help="skip the c3 / c5 sub-lines of the default line (BASELINE configs[2] and [4], 5 steps each)")
    ap.add_argument("--lr", type=float, default=4e-4, help="Adam learning rate of the timed steps (0: the weights never change -- same-data "
                    "A/B runs of timing-only kernel variants, tools/bench_fields.py)")
    args = ap.parse_args()
    conf = CONFIGS[args.config]
    args.dtype = args.dtype or conf["dtype"]
    C2, B_PER_GPU, T = conf["cfg"], conf["B"], conf["T"]      # (the names the formulas below were written with)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("WAE_BENCH_SHARED_GPU", "0") != "0":
            # plumbing check of the N > 1 path on a box with ONE GPU: every rank on cuda:0, collectives over gloo.  The line it
            # prints is labelled as such and is not a measurement of anything.
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    from oracle import wae_oracle as O          # closed-form weights + the cpu_baseline leg only
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine

    geom = Geometry.from_cfg(C2)
    sd = O.make_state_dict(dict(C2), salt=conf["salt"], with_encoder=conf["encoder"])
    eng = WaeEngine(geom, dtype=args.dtype, device=str(device))
    eng.load_state_dict(sd)
    x, lat, g = synth_inputs(rank, device, conf)
    fwd_fn = eng.forward if conf["encoder"] else eng.decoder_forward
    lengths = torch.full((B_PER_GPU,), T, dtype=torch.int32, device=device)
    xi = x.to(torch.int32)

    ev, ev_tn = [], []
    gsync = None
    if args.mode == "train":
        eng.init_optimizer()
        if dist is not None:
            from wavenet_autoencoders_amd.distributed import GradSync, broadcast_params
            broadcast_params(eng.params)
            gsync = GradSync(eng, timing=True)

    def step(record=False):
        if args.mode == "forward":
            # (weight norm + fragment packing run inside decoder_forward only when a parameter changed since the last pack: never here)
            out = fwd_fn(xi, lat, g, targets=xi, lengths=lengths, want_logits=False, layer_events=ev if record else None)
            return out["loss"]
        eng._layer_events = ev if record is True else None
        eng._tn_events = ev_tn if record is True else None
        eng._tm_events = ev_tm if record == "tm" else None      # (only in the untimed extra steps below: 96 event records per step)
        return eng.train_step(xi, lat, g, lengths=None, lr=args.lr, grad_sync=gsync)["loss"]

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ev_tm = {"gate": [], "res": [], "pair": []}
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(record=True)
    sync()
    dt = time.perf_counter() - t0
    if args.mode == "train":
        # the per-layer backward launches (gate / residual: 48 per step) get their HIP events in three EXTRA, untimed steps: an event
        # pair around every launch costs the timed region ~0.4 ms per step (measured: 5.9 -> 6.3 ms), two pairs per step cost nothing
        for _ in range(3):
            step(record="tm")
        sync()
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss_v = float(loss)

    # dominant kernel: the fused layer kernel (glu_kernel_name) -- HIP events recorded on the launch stream around the 24-layer stack of
    # every timed step; average per launch (includes the inter-kernel gaps, so it is conservative)
    stack_ms = [a.elapsed_time(b) for a, b in ev]
    glu_ms = sum(stack_ms) / len(stack_ms) / geom.layers
    es = 4 if args.dtype == "fp32" else 2
    samples = B_PER_GPU * T
    bytes_per_launch = (2 * C2["R"] + 2 * C2["S"] + C2["Cc"]) * es * samples        # SURVEY 8(d) per layer
    H = C2["G"] // 2
    flops_per_launch = 2 * (C2["G"] * C2["R"] * C2["k"] + C2["G"] * C2["Cc"] + H * C2["R"] + H * C2["S"]) * samples
    achieved_gbs = bytes_per_launch / (glu_ms * 1e-3) / 1e9
    achieved_tf = flops_per_launch / (glu_ms * 1e-3) / 1e12
    peak_tf = FP32_MFMA_PEAK_TF if args.dtype == "fp32" else MFMA_PEAK_TF      # fp16 and bf16 MFMA run at the same dense rate
    fwd_roof = {"bound": "hbm", "kernel": glu_kernel_name(args.dtype, args.mode == "train"), "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": glu_ms,
                "algorithmic_bytes_per_launch": bytes_per_launch, "mfma_achieved_tflops": achieved_tf,
                "mfma_frac": achieved_tf / peak_tf,
                "note": "SURVEY 8(d) forward bytes per layer (2R+2S+Cc)*e x %d samples; HIP events around the %d-layer stack" % (samples, geom.layers)}
    roof = fwd_roof
    extra = {}
    if args.mode == "train":
        # Training launch of the fused layer kernel: it also saves the pre-activations -- SURVEY 8(d) train bytes, forward part =
        # (2R + 2S + Cc) + G per layer and sample.
        tb = (2 * C2["R"] + 2 * C2["S"] + C2["Cc"] + C2["G"]) * es * samples
        glu_roof = dict(fwd_roof, achieved=tb / (glu_ms * 1e-3) / 1e9, frac=tb / (glu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        algorithmic_bytes_per_launch=tb, ms_per_step=glu_ms * geom.layers,
                        note="training forward of one layer: SURVEY 8(d) (2R+2S+Cc+G)*e x %d samples (z saved for backward); "
                             "HIP events around the %d-layer stack" % (samples, geom.layers))
        fplan = eng.chain_plan(B_PER_GPU, T)
        if fplan is not None:
            # two chains (engine.chain_plan): a launch covers its chain's clips; the chains run in step, so a launch lasts what a layer
            # of the stack takes (rocprofv3's average for the kernel agrees).  `achieved` = the bytes ONE launch moves over that time;
            # `both_chains` = the layer's bytes of both chains over the same time.
            fpart = fplan[0] / B_PER_GPU
            glu_roof = dict(glu_roof, achieved=glu_roof["achieved"] * fpart, frac=glu_roof["frac"] * fpart,
                            algorithmic_bytes_per_launch=tb * fpart, mfma_achieved_tflops=achieved_tf * fpart, mfma_frac=achieved_tf * fpart / peak_tf,
                            clips_per_launch=fplan[0], concurrent_launches=2,
                            both_chains={"achieved": tb / (glu_ms * 1e-3) / 1e9, "frac": tb / (glu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "ms_per_layer": glu_ms,
                                         "note": "two half-batch chains of launches on two streams, in step: the layer's bytes of both chains / "
                                                 "(wall time of the stack / its layers)"})
        families = {"glu_fwd_z": glu_roof}
        if ev_tn:
            # weight gradients: 16-bit = ONE launch for all layers (gemm_tn_static_kernel: dW1 taps, dWc + zb sums, dW_out AND dW_skip
            # + the out bias), fp32 = one gemm_tn_kernel per layer.  Algorithmic bytes = the operands each layer must read once: dz
            # (G), x (R), c (Cc), dx-hat (R), u (H); dS (S) is the same array for every layer: once per launch.
            # launches per step: 1 (the whole stack), 2 (data parallel: the stack in two layer halves, backward.decoder_backward) or one
            # per layer (fp32 tiles).  The halves are summed: `avg_launch_ms` then is the weight-gradient time of a step.
            per_step = max(1, len(ev_tn) // args.steps)
            nl = geom.layers if per_step <= 2 else 1
            tn_ms = sum(a.elapsed_time(b) for a, b in ev_tn) / len(ev_tn) * (per_step if nl > 1 else 1)
            from wavenet_autoencoders_amd import backward as BW
            bws = eng._ws[("bwd", B_PER_GPU, T)]
            static = nl > 1 and isinstance(bws["stream"], BW.StaticStreamTable)
            beside = None
            if "stream_beside" in bws and per_step == 2 and dist is None:
                # an under-filled sweep (backward.decoder_backward: `beside`): two launches per step, the upper layers' on the idle CUs BESIDE
                # the sweep (its duration is mostly hidden, and long: a third of the machine), the lower layers' behind the sweep
                t_b = sum(a.elapsed_time(b) for a, b in ev_tn[0::2]) / len(ev_tn[0::2])
                t_l = sum(a.elapsed_time(b) for a, b in ev_tn[1::2]) / len(ev_tn[1::2])
                beside = {"beside_sweep_launch_ms": t_b, "behind_sweep_launch_ms": t_l, "sum_ms": tn_ms,
                          "note": "two launches of the same kernel: the upper layers' + the head's sized for the CUs the sweep leaves idle, on a side "
                                  "stream beside the sweep's lower part; the lower layers' behind the sweep.  avg_launch_ms = their sum "
                                  "(the work), of which only ~behind_sweep_launch_ms extends the step"}
            tn_bytes = (nl * (C2["G"] + 2 * C2["R"] + C2["Cc"] + H) + (C2["S"] if static else 0)) * es * samples
            tn_flops = nl * 2 * (C2["G"] * (C2["R"] * C2["k"] + C2["Cc"]) + C2["R"] * H + (C2["S"] * H if static else 0)) * samples
            tn_gbs = tn_bytes / (tn_ms * 1e-3) / 1e9
            families["wgrad"] = extra["roofline_wgrad"] = {
                "bound": "hbm", "kernel": ("gemm_tn_static_kernel" if static else "gemm_tn_stream_kernel") if nl > 1 else "gemm_tn_kernel",
                "achieved": tn_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tn_gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": tn_ms,
                "ms_per_step": tn_ms * (1 if nl > 1 else geom.layers),
                "algorithmic_bytes_per_launch": tn_bytes, "mfma_achieved_tflops": tn_flops / (tn_ms * 1e-3) / 1e12,
                "mfma_frac": tn_flops / (tn_ms * 1e-3) / 1e12 / peak_tf,
                "note": "weight gradients of %d layer(s) per launch (dW1 taps, dWc, dW_out%s); operands read once = (G+2R+Cc+H)*e per "
                        "sample and layer%s" % (nl, ", dW_skip" if static else "", " + S*e per sample" if static else "")}
            if beside is not None:
                families["wgrad"]["beside_sweep"] = beside
                families["wgrad"]["ms_per_step"] = beside["behind_sweep_launch_ms"]      # (what extends the step: the family ranking uses it)
        # (csrc/glu_bwd8.hip has the instantiations of the 256-wide 16-bit shapes; csrc/glu_bwd.hip the rest.  The two-launch sweep's
        #  residual (mode 1) and gate (mode 2) launches run on csrc/gemm_tm8.hip's gemm_tm8x_kernel in 16-bit storage at the bench shapes.)
        pair_kernel = "glu_bwd_pair8_kernel" if (es == 2 and C2["R"] == 256 and C2["S"] == 256 and H in (192, 128, 184) and C2["k"] == 3) else "glu_bwd_pair_kernel"
        for kind, mode_id, kb, kf, what in (
                ("gate", 2, (C2["R"] + C2["S"] + 2 * C2["G"]) * es, 2 * H * (C2["R"] + C2["S"]),
                 "du/dz of one layer: reads dx-hat (R), dskip (S), the saved pre-activations (G), writes dz (G)"),
                ("res", 1, (C2["G"] + 2 * C2["R"]) * es, 2 * C2["R"] * C2["G"] * C2["k"],
                 "dx-hat of one layer: reads dz (G; three taps of the same rows) and dx-hat of the layer above (R), writes R"),
                ("pair", 0, (3 * C2["G"] + 2 * C2["R"] + C2["S"]) * es, 2 * C2["R"] * C2["G"] * C2["k"] + 2 * H * (C2["R"] + C2["S"]),
                 "residual(l) + gate(l-1) in one launch (csrc/glu_bwd.hip, the 16-bit default): reads dz_l (G), dx-hat of the layer above "
                 "(R), dskip (S), the saved pre-activations of layer l-1 (G), writes dx-hat_l (R) and dz_(l-1) (G)")):
            if ev_tm[kind]:
                k_ms = sum(a.elapsed_time(b) for a, b in ev_tm[kind]) / len(ev_tm[kind])
                per_step = len(ev_tm[kind]) / 3
                # two chains (engine.chain_plan): the events bracket the FIRST chain's launches, each over its own clips only, while the
                # second chain's launch of the same kernel runs beside it.  `achieved` is what the rubric asks for -- the bytes ONE launch
                # moves over ITS duration (rocprofv3's average for the kernel agrees with it) -- and `both_chains` prices the sweep: the
                # bytes of both chains' launches of a layer over the sweep's wall time per layer (HIP events around the whole sweep).
                plan = eng.chain_plan(B_PER_GPU, T, backward=True)
                part = (plan[0] / B_PER_GPU) if (plan is not None and kind != "gate") else 1.0     # (the top layer's gate launch runs before the fork)
                ls = samples * part
                gbs = kb * ls / (k_ms * 1e-3) / 1e9
                fam = {
                    "bound": "hbm", "kernel": (("gemm_tm8x_kernel:%d" if es == 2 else "gemm_tm_kernel:%d") % mode_id) if mode_id else pair_kernel, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": k_ms, "ms_per_step": k_ms * per_step,
                    "algorithmic_bytes_per_launch": kb * ls, "mfma_achieved_tflops": kf * ls / (k_ms * 1e-3) / 1e12,
                    "mfma_frac": kf * ls / (k_ms * 1e-3) / 1e12 / peak_tf,
                    "note": what + " -- the bytes this launch itself has to move; HIP events around every launch"}
                if part < 1.0:
                    fam["note"] = ("TWO launches of this kernel run side by side, each over half of the batch (engine.chain_plan): achieved / frac "
                                   "are ONE launch's bytes over its own duration; `both_chains` prices the pair.  ") + fam["note"]
                if part < 1.0 and ev_tm.get("sweep"):
                    sw_ms = sum(a.elapsed_time(b) for a, b in ev_tm["sweep"]) / len(ev_tm["sweep"])
                    lay_ms = sw_ms / max(per_step, 1)
                    fam["clips_per_launch"] = plan[0]
                    fam["concurrent_launches"] = 2
                    fam["ms_per_step"] = sw_ms
                    fam["both_chains"] = {"achieved": kb * samples / (lay_ms * 1e-3) / 1e9, "frac": kb * samples / (lay_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "ms_per_layer": lay_ms, "sweep_ms": sw_ms,
                                          "note": "two half-batch chains of launches on two streams: bytes of both chains' launches of a layer / "
                                                  "(wall time of the sweep / its layers)"}
                families[kind] = extra["roofline_" + {"gate": "gate_bwd", "res": "residual_bwd", "pair": "bwd_pair"}[kind]] = fam
        # `roofline` = the launch family with the largest share of the timed step (HIP events on the launch stream decide, not a comment)
        top = max(families, key=lambda k: families[k]["ms_per_step"])
        roof = dict(families[top], family=top,
                    family_ms_per_step={k: round(v["ms_per_step"], 4) for k, v in families.items()})
        extra["roofline_glu_fwd_z"] = glu_roof
        # the whole step against SURVEY 8(d): per sample and layer (5R + 3S + 2G + 4Cc)*e train bytes (+ head), 3 x forward FLOPs
        step_bytes = (geom.layers * (5 * C2["R"] + 3 * C2["S"] + 2 * C2["G"] + 4 * C2["Cc"]) * es + (C2["S"] + 0) * es + 5) * samples
        fwd_flops = (geom.layers * 2 * (C2["G"] * C2["R"] * C2["k"] + C2["G"] * C2["Cc"] + H * C2["R"] + H * C2["S"])
                     + 2 * (C2["S"] * C2["S"] + C2["S"] * C2["O"])) * samples
        step_s = dt / args.steps
        extra["roofline_step"] = {"hbm_frac": step_bytes / step_s / 1e9 / HBM_PEAK_GBS, "mfma_frac": 3 * fwd_flops / step_s / 1e12 / peak_tf,
                                  "algorithmic_bytes_per_step": step_bytes, "algorithmic_flops_per_step": 3 * fwd_flops,
                                  "note": "SURVEY 8(d): train bytes (5R+3S+2G+4Cc)*e per sample and layer + head, 3 x forward FLOPs, over ms_per_step"}
    fwd_bytes_per_sample = geom.layers * (2 * C2["R"] + 2 * C2["S"] + C2["Cc"]) * es + (C2["S"] + 0) * es + 4 + 1
    fwd_bytes_whole = geom.layers * (2 * C2["R"] + 2 * C2["S"] + C2["Cc"]) * es + (C2["S"] + C2["O"]) * es + 1      # 53 249 B (bf16)
    value = world * samples * args.steps / dt

    # HBM traffic per launch: from the committed rocprofv3 --pmc passes of this same command (bench.py cannot run the profiler
    # on itself).  The file carries the hash of the kernel sources it was measured on: a different build reports null.
    traffic_src, traffic_doc = None, None
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
        for fpath in reversed(cands):
            with open(fpath) as fh:
                doc = json.load(fh)
            if doc.get("csrc_hash") == csrc_hash():
                traffic_src = os.path.basename(fpath)
                traffic_doc = doc
                # (the passes profile the DEFAULT command: the C2 shard in bf16 -- another configuration's launches move other bytes)
                if args.dtype == "bf16" and args.mode == "train" and args.config == "c2":
                    for rf in [roof] + [v for v in extra.values() if "kernel" in v]:
                        for k, v in doc["kernels"].items():
                            if kernel_matches(rf["kernel"], k):
                                rf["traffic"] = v["hbm_bytes_per_launch"]
                                rf["traffic_source"] = traffic_src
                break
    except (OSError, KeyError, ValueError):
        pass
    if traffic_src is not None and roof.get("traffic") is None:
        roof["traffic_note"] = "%s profiles the default command (config c2, bf16, train): no PMC pass of this configuration" % traffic_src
    if traffic_src is None:
        roof["traffic_note"] = "no profiles/*_pmc_traffic.json matches this build's csrc hash %s: traffic not reported" % csrc_hash()

    if rank == 0:
        res = {
            "metric": "teacher-forced audio samples/sec (%d-layer decoder), " % geom.layers + ("train step" if args.mode == "train" else "forward + CE"),
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "dtype_parity": {"fp32": "logits / latents / gradients within 1e-3 of the reference (measured ~1e-6): the north star's tolerance",
                             "bf16": "16-bit storage, fp32 accumulate: checked at max|a-b| / max|b| <= 5e-2 per tensor against the fp32 oracle, "
                                     "VQ indices bit-exact (tests/helpers.py, DESIGN.md section 4)",
                             "fp16": "as bf16 at 1e-2 (logits) / 4e-2 (gradients), loss scale 4096"}[args.dtype],
            "config": {"workload": conf["name"] + f", batch {B_PER_GPU}x{T} per GPU, "
                                   + ("full train step: weight-norm+pack, " + ("encoder, VQ, " if conf["encoder"] else "") + "forward, fused CE, backward, "
                                      "clip+Adam+EMA" if args.mode == "train" else "teacher-forced forward (weights packed once, before the timed region), upsample, head and fused CE")
                                   + "; closed-form random weights", "name": args.config,
                       "global_batch": world * B_PER_GPU, "seq_len": T, "parallelism": f"dp{world}"},
            "samples_per_sec_per_gpu": value / world,
            "loss": loss_v,
            "roofline": roof,
            "roofline_glu_fwd": fwd_roof,
        }
        res.update(extra)
        if world > 1 and os.environ.get("WAE_BENCH_SHARED_GPU", "0") != "0":
            res["data"] = "synthetic; PLUMBING CHECK: %d ranks share ONE GPU over gloo -- not a measurement" % world
        if gsync is not None:
            comm_ms, wait_ms = gsync.collect_timing()
            res["allreduce"] = {"ms": comm_ms, "exposed_ms": wait_ms, "collectives_per_step": gsync.n_collectives / (args.steps + args.warmup),
                                "bytes": int(eng.grads.numel()) * (2 if gsync.wire_bf16 else 4), "wire_dtype": "bf16" if gsync.wire_bf16 else "fp32",
                                "note": "RCCL all-reduce of the fp32 gradient arena (WAE_DP_WIRE=bf16: of bf16 copies), last timed step: "
                                "ms = first launch to last collective done on the side stream (starts inside backward, after the "
                                "layers' gradients); exposed_ms = how long the compute stream waited for it"}
        if args.mode == "train" and world == 1:
            # the same stack in inference (no z saved, no backward): the north star states its roofline target on this launch
            ev_f = []
            nf = max(3, min(10, args.steps))
            # weight norm + packing run when the parameters change (WaeEngine tracks their generation), not per pass: an inference
            # pass over unchanged weights is the reference after make_generation_fast_ (wavenet.py:358-364).  The TRAINING forward
            # above recomputes them every step, like the reference's weight-norm hooks.
            eng.prepare_weights()
            for i in range(2 + nf):
                fwd_fn(xi, lat, g, targets=xi, lengths=lengths, want_logits=False, layer_events=ev_f if i >= 2 else None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(nf):
                fwd_fn(xi, lat, g, targets=xi, lengths=lengths, want_logits=False)
            e1.record()
            # ... and the same pass INCLUDING weight norm + fragment packing (what a forward cost in this line up to round 3, and what
            # the reference's forward pays while its weight-norm hooks are attached: every pass re-derives w from g, v)
            e2 = torch.cuda.Event(enable_timing=True)
            for _ in range(nf):
                eng.prepare_weights()
                fwd_fn(xi, lat, g, targets=xi, lengths=lengths, want_logits=False)
            e2.record()
            torch.cuda.synchronize()
            f_ms = sum(a.elapsed_time(b) for a, b in ev_f) / len(ev_f) / geom.layers
            f_step = e0.elapsed_time(e1) / nf
            f_step_prep = e1.elapsed_time(e2) / nf
            res["forward_inference"] = {
                "metric": "teacher-forced audio samples/sec (%d-layer decoder), forward + CE" % geom.layers, "value": samples / (f_step * 1e-3),
                "unit": "samples/s", "ms_per_step": f_step, "steps": nf,
                "ms_per_step_with_weight_prep": f_step_prep,
                "weight_prep_note": "ms_per_step: weight norm + fragment packing done once before the timed passes (inference over fixed "
                                    "weights, the reference after make_generation_fast_); ms_per_step_with_weight_prep: redone in every "
                                    "pass (the definition of this leg up to round 3; BASELINE's 30 % target is stated on the layer stack and "
                                    "priced here on ms_per_step)",
                "roofline_whole": {"bound": "hbm", "achieved": fwd_bytes_whole * samples / (f_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": fwd_bytes_whole * samples / (f_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "algorithmic_bytes_per_sample": fwd_bytes_whole,
                                   "note": "the whole inference forward (%supsample, speaker projection, first conv, %d layers, head, fused CE; "
                                           "weights prepared once) on SURVEY 8(d)'s DECODER forward bytes per sample; BASELINE.md derives the "
                                           "30 %% target (C2: <= 1.42 ms) from this figure" % ("encoder, VQ, " if conf["encoder"] else "", geom.layers)},
                "roofline": {"bound": "hbm", "kernel": glu_kernel_name(args.dtype, False), "achieved": bytes_per_launch / (f_ms * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_per_launch / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": None, "avg_launch_ms": f_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                             "mfma_frac": flops_per_launch / (f_ms * 1e-3) / 1e12 / peak_tf,
                             "note": "inference launch (no z saved): SURVEY 8(d) forward bytes (2R+2S+Cc)*e x %d samples; HIP "
                                     "events around the %d-layer stack" % (samples, geom.layers)}}
            if traffic_doc is not None and args.dtype in ("bf16", "fp16"):
                # the inference launch is its own kernel symbol: its own PMC entry (the train-mode profile runs this leg too)
                for k, v in traffic_doc["kernels"].items():
                    if kernel_matches(res["forward_inference"]["roofline"]["kernel"], k):
                        res["forward_inference"]["roofline"]["traffic"] = v["hbm_bytes_per_launch"]
                        res["forward_inference"]["roofline"]["traffic_source"] = traffic_src
        if args.config == "c2" and args.dtype != "fp32" and args.mode == "train" and world == 1 and not args.no_fp32:
            res["fp32_parity_mode"] = fp32_leg(conf, sd, device)
        if not args.no_ar and world == 1 and args.config == "c2":
            res["autoregressive"] = ar_leg(device, cpu=not args.no_cpu, full_clip=not args.ar_short)
        if not args.no_cpu and world == 1:
            # a bounded sample (~10-30 s of host work): c2 / c3 the whole shard, c5 (105 MFLOP per sample forward) two clips
            ncl = 2 if args.config == "c5" else B_PER_GPU
            cb, cpu_loss = cpu_baseline_train(sd, ncl, conf) if args.mode == "train" else cpu_baseline(sd, ncl, conf)
            res["cpu_baseline"] = cb
        if args.config == "c2" and args.mode == "train" and world == 1 and not args.no_sub and args.dtype == conf["dtype"]:
            # BASELINE configs[2] and [4] -- the two 8-GPU configurations -- as their per-GPU shard on this GPU: a few steps each, run as
            # CHILD processes of this one (never an exec from a process that holds the GPU) after everything above has been timed.  The
            # driver's line then carries all three trainable configurations; the full lines are `bench.py --config c3|c5`.
            del eng
            torch.cuda.empty_cache()
            for sub in ("c3", "c5"):
                res[sub] = sub_config_line(sub, cpu=not args.no_cpu)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

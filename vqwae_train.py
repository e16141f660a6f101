#!/usr/bin/env python
"""Trainining script for VQ-WaveNet autoencoders on MI355X (entry point and flags of the reference's vqwae_train.py:1-18).

usage: vqwae_train.py [options]

options:
    --dump-root=<dir>            Directory contains preprocessed features.
    --checkpoint-dir=<dir>       Directory where to save model checkpoints [default: checkpoints].
    --hparams=<parmas>           Hyper parameters [default: ].
    --preset=<json>              Path of preset parameters (json).
    --checkpoint=<path>          Restore model from checkpoint path if given.
    --restore-parts=<path>       Restore part of the model.
    --log-event-path=<name>      Log event path.
    --reset-optimizer            Reset optimizer.
    --use-norm=<bool>            Use normalised features [default: true].
    --feat=<name>                Feature file stem [default: mfcc].
    --speaker-id=<N>             Ignored, as in the reference (vqwae_train.py:1072-1073).
    --dtype=<fp32|bf16>          Compute precision of the decoder stack [default: bf16].
    --synthetic                  Train on synthetic batches (no dataset needed).
    --max-steps=<N>              Stop after N steps (overrides max_train_steps).

One process per GPU: launch with ``python -m torch.distributed.run --nproc-per-node N vqwae_train.py ...`` for data
parallel training (gradients are all-reduced over RCCL; every rank reads its own shard).
"""
import argparse
import json
import os
import shutil
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from wavenet_autoencoders_amd import Geometry, lrschedule  # noqa: E402
from wavenet_autoencoders_amd import distributed as D  # noqa: E402
from wavenet_autoencoders_amd.data import CropBatcher, SyntheticBatcher, read_index  # noqa: E402
from wavenet_autoencoders_amd.hparams import hparams  # noqa: E402


def build_geometry(hp) -> Geometry:
    """build_model (vqwae_train.py:913-947): WaveNet from hparams wrapped in VQVAE(c_in=dim_in, hid=cin_channels,
    encoder_hid); note the reference never forwards hparams.K, so the codebook always has 256 entries (:946)."""
    from wavenet_autoencoders_amd.wavenet_vocoder.util import is_mulaw_quantize, is_scalar_input
    if is_mulaw_quantize(hp.input_type) and hp.out_channels != hp.quantize_channels:
        raise RuntimeError("out_channels must equal to quantize_chennels if input_type is 'mulaw-quantize'")
    return Geometry(layers=hp.layers, stacks=hp.stacks, R=hp.residual_channels, G=hp.gate_channels, S=hp.skip_out_channels,
                    O=hp.out_channels, Cc=hp.cin_channels, Cg=hp.gin_channels, k=hp.kernel_size, n_speakers=hp.n_speakers,
                    upsample_scales=list(hp.upsample_params["upsample_scales"]) if hp.upsample_conditional_features else None,
                    cin_pad=hp.cin_pad, scalar_input=is_scalar_input(hp.input_type), use_speaker_embedding=True,
                    c_in=hp.dim_in, encoder_hid=hp.encoder_hid, K=256)


def save_checkpoint(eng, step, epoch, checkpoint_dir, hp, rank):
    """Reference layout (vqwae_train.py:878-910): {"state_dict","optimizer","global_step","global_epoch",
    "global_test_step"} -> checkpoint_step{:09d}.pth + checkpoint_latest.pth, and an _ema twin with the shadow weights."""
    if rank != 0:
        return
    os.makedirs(checkpoint_dir, exist_ok=True)
    path = os.path.join(checkpoint_dir, "checkpoint_step{:09d}.pth".format(step))
    sd = {k: v.cpu() for k, v in eng.state_dict().items()}
    opt = None
    if hp.save_optimizer_state and hasattr(eng, "exp_avg"):
        opt = dict(exp_avg=eng.exp_avg.cpu(), exp_avg_sq=eng.exp_avg_sq.cpu(), step=eng.opt_step, layout="flat-arena")
    torch.save({"state_dict": sd, "optimizer": opt, "global_step": step, "global_epoch": epoch, "global_test_step": 0}, path)
    shutil.copyfile(path, os.path.join(checkpoint_dir, "checkpoint_latest.pth"))
    if getattr(eng, "shadow", None) is not None:
        ema_sd = {}
        for k in eng.lay.offsets:
            off, n = eng.lay.off(k), eng.lay.numel(k)
            ema_sd[k] = eng.shadow[off:off + n].view(eng.lay.shapes[k]).cpu()
        epath = os.path.join(checkpoint_dir, "checkpoint_step{:09d}_ema.pth".format(step))
        torch.save({"state_dict": ema_sd, "optimizer": None, "global_step": step, "global_epoch": epoch, "global_test_step": 0}, epath)
        shutil.copyfile(epath, os.path.join(checkpoint_dir, "checkpoint_latest_ema.pth"))
    print("Saved checkpoint:", path)


def load_checkpoint(path, eng, reset_optimizer):
    """vqwae_train.py:959-976"""
    ck = torch.load(path, map_location="cpu")
    eng.load_state_dict(ck["state_dict"])
    eng.init_optimizer()
    opt = ck.get("optimizer")
    if not reset_optimizer and isinstance(opt, dict) and opt.get("layout") == "flat-arena":
        eng.exp_avg.copy_(opt["exp_avg"])
        eng.exp_avg_sq.copy_(opt["exp_avg_sq"])
        eng.opt_step = int(opt["step"])
    return int(ck.get("global_step", 0)), int(ck.get("global_epoch", 0))


def restore_parts(path, eng):
    """load matching keys only (vqwae_train.py:980-999)"""
    sd = torch.load(path, map_location="cpu")["state_dict"]
    cur = eng.state_dict()
    for k, v in sd.items():
        if k in cur and tuple(cur[k].shape) == tuple(v.shape):
            cur[k] = v
        else:
            print("warn: skip", k)
    eng.load_state_dict({k: v.cpu() for k, v in cur.items()})


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--dump-root")
    ap.add_argument("--checkpoint-dir", default="checkpoints")
    ap.add_argument("--hparams", default="")
    ap.add_argument("--preset")
    ap.add_argument("--checkpoint")
    ap.add_argument("--restore-parts")
    ap.add_argument("--log-event-path")
    ap.add_argument("--reset-optimizer", action="store_true")
    ap.add_argument("--use-norm", default="true")
    ap.add_argument("--feat", default="mfcc")
    ap.add_argument("--speaker-id")
    ap.add_argument("--dtype", default="bf16", choices=["fp32", "bf16"])
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--max-steps", type=int)
    args = ap.parse_args(argv)

    if args.preset:
        with open(args.preset) as f:
            hparams.parse_json(f.read())
    hparams.parse(args.hparams)
    hp = hparams
    rank, local, world = D.init_from_env()
    device = f"cuda:{local}"
    torch.cuda.set_device(device)
    os.makedirs(args.checkpoint_dir, exist_ok=True)
    if rank == 0:
        with open(os.path.join(args.checkpoint_dir, "hparams.json"), "w") as f:       # vqwae_train.py:1100-1102
            json.dump(hp.values(), f, indent=2)

    from wavenet_autoencoders_amd.engine import WaeEngine
    geom = build_geometry(hp)
    eng = WaeEngine(geom, dtype=args.dtype, device=device)
    # reference initialisation, identical on every rank
    torch.manual_seed(1234)
    from wavenet_autoencoders_amd.wavenet_vocoder._base import ArenaModel, register_params

    class _Init(ArenaModel):
        pass
    tmp = _Init()
    tmp._init_arena(geom, "")
    eng.load_state_dict({k: v.detach() for k, v in tmp.state_dict().items()})
    step, epoch = 0, 0
    if args.restore_parts:
        restore_parts(args.restore_parts, eng)
    if args.checkpoint:
        step, epoch = load_checkpoint(args.checkpoint, eng, args.reset_optimizer)
    D.broadcast_params(eng.params)
    eng.init_optimizer(ema=bool(hp.exponential_moving_average)) if not hasattr(eng, "exp_avg") else None

    if hp.batch_size % world != 0:
        raise ValueError("batch size % num gpu must be 0 (vqwae_train.py:754)")
    per_rank = hp.batch_size // world
    hop = hp.hop_size
    if args.synthetic or not args.dump_root:
        loader = SyntheticBatcher(per_rank, hop, hp.max_time_steps, hp.dim_in, hp.n_speakers, steps=args.max_steps or 20, rank=rank)
    else:
        feat = args.feat + (".norm.npy" if str(args.use_norm).lower() in ("true", "1") else ".npy")
        items = read_index(args.dump_root, "train_no_dev", hp.max_time_steps // hop)
        loader = CropBatcher(items, per_rank, hop, hp.max_time_steps, feat, hp.cin_pad, rank, world)
    bucketer = {}

    def grad_hook(grads):
        if world > 1:
            if "b" not in bucketer:
                bucketer["b"] = D.GradBucketer(grads)
            bucketer["b"].finish()

    max_steps = args.max_steps or hp.max_train_steps
    sched = getattr(lrschedule, hp.lr_schedule) if hp.lr_schedule else None
    t0 = time.time()
    try:
        while epoch < hp.nepochs and step < max_steps:
            for x, c, g, lengths in loader:
                lr = hp.optimizer_params["lr"]
                if sched is not None:
                    lr = sched(lr, step, **hp.lr_schedule_kwargs)                     # vqwae_train.py:729-735
                res = eng.train_step(x.to(device), c.to(device), g.to(device), lengths=None if bool((lengths == x.shape[1]).all())
                                     else lengths, lr=lr, eps=hp.optimizer_params.get("eps", 1e-8),
                                     weight_decay=hp.optimizer_params.get("weight_decay", 0.0), clip_thresh=hp.clip_thresh,
                                     ema_decay=hp.ema_decay, grad_hook=grad_hook)
                step += 1
                if step % 10 == 0 or step == 1:
                    stats = torch.stack([res["loss"].float(), res.get("vq_loss", res["loss"]).float(),
                                         res.get("perp", res["loss"]).float()])
                    D.all_reduce_scalars(stats)
                    if rank == 0:
                        dt = (time.time() - t0) / step
                        print(f"step {step} loss {float(stats[0]):.4f} vq {float(stats[1]):.4f} perp {float(stats[2]):.2f} "
                              f"gnorm {float(res['grad_norm']):.3f} lr {lr:.2e} {hp.batch_size * x.shape[1] / dt / 1e6:.2f} Msamples/s")
                if step % hp.checkpoint_interval == 0:
                    save_checkpoint(eng, step, epoch, args.checkpoint_dir, hp, rank)
                if step >= max_steps:
                    break
            epoch += 1
    except KeyboardInterrupt:
        pass
    finally:
        save_checkpoint(eng, step, epoch, args.checkpoint_dir, hp, rank)             # vqwae_train.py:1140-1145
    if rank == 0:
        print("Finished")
    return 0


if __name__ == "__main__":
    sys.exit(main())

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
    --dtype=<fp32|bf16|fp16>     Storage precision of the decoder stack [default: bf16].
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
from wavenet_autoencoders_amd.checkpoint import load_checkpoint, restore_parts, save_checkpoint  # noqa: E402
from wavenet_autoencoders_amd.data import CropBatcher, Prefetcher, SyntheticBatcher, read_index  # noqa: E402
from wavenet_autoencoders_amd.hparams import adam_settings, hparams  # noqa: E402


def build_geometry(hp) -> Geometry:
    """build_model (vqwae_train.py:913-947): WaveNet from hparams wrapped in VQVAE(c_in=dim_in, hid=cin_channels,
    encoder_hid); note the reference never forwards hparams.K, so the codebook always has 256 entries (:946)."""
    from wavenet_autoencoders_amd.wavenet_vocoder.util import is_mulaw_quantize, is_scalar_input
    if is_mulaw_quantize(hp.input_type) and hp.out_channels != hp.quantize_channels:
        raise RuntimeError("out_channels must equal to quantize_chennels if input_type is 'mulaw-quantize'")
    if hp.upsample_conditional_features and hp.upsample_net not in ("ConvInUpsampleNetwork", "UpsampleNetwork"):
        raise AttributeError(f"module 'wavenet_vocoder.upsample' has no attribute {hp.upsample_net!r}")       # wavenet.py:150
    up_act, up_slope = "none", 0.01
    if hp.upsample_conditional_features:
        from wavenet_autoencoders_amd.packing import UP_ACT_KINDS
        up_act = hp.upsample_params.get("upsample_activation", "none")
        ap = {k: v for k, v in dict(hp.upsample_params.get("upsample_activation_params", {}) or {}).items() if k != "inplace"}
        if up_act == "LeakyReLU":
            up_slope = float(ap.pop("negative_slope", 0.01))
        if up_act != "none" and (up_act not in UP_ACT_KINDS or ap):
            raise NotImplementedError(f"upsample_params.upsample_activation={up_act!r} {ap}: ReLU, LeakyReLU, Tanh, Sigmoid are implemented")
        for key, ok in (("mode", "nearest"), ("freq_axis_kernel_size", 1)):
            if hp.upsample_params.get(key, ok) != ok:
                raise NotImplementedError(f"upsample_params.{key}={hp.upsample_params[key]!r} is not implemented (only {ok!r})")
    return Geometry(layers=hp.layers, stacks=hp.stacks, R=hp.residual_channels, G=hp.gate_channels, S=hp.skip_out_channels,
                    O=hp.out_channels, Cc=hp.cin_channels, Cg=hp.gin_channels, k=hp.kernel_size, n_speakers=hp.n_speakers,
                    upsample_scales=list(hp.upsample_params["upsample_scales"]) if hp.upsample_conditional_features else None,
                    cin_pad=hp.cin_pad, scalar_input=is_scalar_input(hp.input_type), use_speaker_embedding=True,
                    c_in=hp.dim_in, encoder_hid=hp.encoder_hid, K=256, conv_in=hp.upsample_net != "UpsampleNetwork",
                    up_act=up_act, up_act_slope=up_slope)


def evaluate(eng, loader, device, hp):
    """The dev phase of train_loop (vqwae_train.py:829-870 with train=False: model.eval(), forward and loss only, no update):
    -> (loss, vq_loss, perplexity) averaged over the phase's batches as the reference's per-epoch log does."""
    tot, n = torch.zeros(3, dtype=torch.float64), 0
    for x, c, g, lengths in loader:
        T = x.shape[1]
        ln = None if bool((lengths == T).all()) else lengths
        if eng.g.scalar_input:
            out = eng.forward(x.to(device), c.to(device), g.to(device), want_logits=True, train=False)
            ce, _ = eng.dmol_loss_and_grad(out["logits"], x.to(device), ln, hp.quantize_channels, hp.log_scale_min)
        else:
            out = eng.forward(x.to(device), c.to(device), g.to(device), targets=x.to(device), lengths=ln, want_logits=False, train=False)
            ce = out["loss"]
        tot += torch.stack([ce.float() + out["vq_loss"].float(), out["vq_loss"].float(), out["perp"].float()]).double().cpu()
        n += 1
    stats = (tot / max(n, 1)).float()
    if device != "cpu":
        stats = stats.to(device)
    D.all_reduce_scalars(stats)
    return [float(v) for v in stats.cpu()]


def eval_model(eng, x, c, g, lengths, global_step, eval_dir, hp, use_ema, hop):
    """eval_model (vqwae_train.py:572-640): online decoding of ONE random item of the batch with the AVERAGED weights when an EMA is
    kept (clone_as_averaged_model, :353-360: the shadow replaces the parameters for the duration of the decode; the reference also
    strips weight norm with make_generation_fast_, which changes nothing numerically): encoder + VQ on the item's features,
    autoregressive decode of `length` samples from the silence class (mulaw_quantize(0, 255) = 127) by categorical sampling
    (softmax=True, quantize=True), then predicted and target waveforms as 16-bit wav files named as the reference names them.
    (The reference's wave plots are out of scope.)  Returns (predicted, target) float waveforms."""
    import numpy as np
    from scipy.io import wavfile
    from wavenet_autoencoders_amd.data import inv_mulaw_quantize
    idx = int(np.random.randint(0, x.shape[0]))
    length = int(lengths[idx])
    y_target = x[idx, :length].detach().cpu().numpy()
    frames = length // hop + 2 * hp.cin_pad if eng.g.upsample_scales else length
    ci = c[idx:idx + 1, :, :frames].contiguous().float()
    gi = g[idx:idx + 1].contiguous() if g is not None else None
    saved = None
    if use_ema and getattr(eng, "shadow", None) is not None:
        print("Using averaged model for evaluation")
        saved = eng.params.clone()
        eng.params.copy_(eng.shadow)
        eng.weights_dirty = True
    try:
        eng.prepare_weights()
        lat = eng.encoder_forward(ci)
        quant, _, _ = eng.vq_forward(lat)
        Tgen = length
        if eng.g.upsample_scales:        # the decoder emits whole latent frames (the reference asserts the same, wavenet.py:198-200)
            Tgen = min(length, (quant.shape[-1] - 2 * hp.cin_pad) * int(np.prod(eng.g.upsample_scales)))
            y_target = y_target[:Tgen]
        if eng.g.scalar_input:
            out = eng.incremental_forward(quant, gi, Tgen, mode="sample", log_scale_min=hp.log_scale_min)
            y_hat = out["x"][0].float().cpu().numpy()
            y_target = y_target.astype(np.float32)
        else:
            out = eng.incremental_forward(quant, gi, Tgen, mode="sample", init_idx=127)
            y_hat = inv_mulaw_quantize(out["idx"][0].cpu().numpy(), hp.quantize_channels - 1)       # :621-623 use 255
            y_target = inv_mulaw_quantize(y_target, hp.quantize_channels - 1)
        eng.check_errors()
    finally:
        if saved is not None:
            eng.params.copy_(saved)
            eng.weights_dirty = True
    os.makedirs(eval_dir, exist_ok=True)
    for tag, y in (("predicted", y_hat), ("target", y_target)):
        pcm = np.clip(np.asarray(y, dtype=np.float64) * 32767.0, -32768, 32767).astype(np.int16)
        wavfile.write(os.path.join(eval_dir, "step{:09d}_{}.wav".format(global_step, tag)), hp.sample_rate, pcm)
    return y_hat, y_target


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
    ap.add_argument("--dtype", default="bf16", choices=["fp32", "bf16", "fp16"])
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
    eng = WaeEngine(geom, dtype=args.dtype, device=device, dropout=float(hp.dropout),            # vqwae_train.py:933
                    drop_seed=0x5EED * 4099 + rank)   # replicas draw their own masks
    # reference initialisation, identical on every rank
    torch.manual_seed(1234)
    from wavenet_autoencoders_amd.wavenet_vocoder._base import ArenaModel, register_params

    class _Init(ArenaModel):
        pass
    tmp = _Init()
    tmp._init_arena(geom, "")
    eng.load_state_dict({k: v.detach() for k, v in tmp.state_dict().items()})
    step, epoch, test_step = 0, 0, 0
    use_ema = bool(hp.exponential_moving_average)
    if args.restore_parts:
        restore_parts(args.restore_parts, eng)
    if args.checkpoint:
        step, epoch, test_step = load_checkpoint(args.checkpoint, eng, args.reset_optimizer, ema=use_ema)
    eng.drop_calls = step        # the dropout masks follow the global step: a resumed run does not replay the first steps' masks
    D.broadcast_params(eng.params)
    if not hasattr(eng, "exp_avg"):
        eng.init_optimizer(ema=use_ema)
    elif use_ema:
        eng.shadow.copy_(eng.params)                 # the shadow registers the (broadcast) weights it starts from (:822-826)

    if hp.batch_size % world != 0:
        raise ValueError("batch size % num gpu must be 0 (vqwae_train.py:754)")
    per_rank = hp.batch_size // world
    hop = hp.hop_size
    dev_loader = None
    if args.synthetic or not args.dump_root:
        loader = SyntheticBatcher(per_rank, hop, hp.max_time_steps, hp.dim_in, hp.n_speakers, steps=args.max_steps or 20, rank=rank)
    else:
        feat = args.feat + (".norm.npy" if str(args.use_norm).lower() in ("true", "1") else ".npy")
        min_frames = (hp.max_time_steps // hop + 2 * hp.cin_pad) if hp.max_time_steps is not None else 0
        items = read_index(args.dump_root, "train_no_dev", min_frames, hp.n_speakers)
        loader = CropBatcher(items, hp.batch_size, hop, hp.max_time_steps, feat, hp.cin_pad, rank, world, train=True,
                             n_classes=hp.quantize_channels)
        if os.path.exists(os.path.join(args.dump_root, "dev", "train.txt")):
            dev_items = read_index(args.dump_root, "dev", min_frames, hp.n_speakers)
            if dev_items:
                dev_loader = CropBatcher(dev_items, hp.batch_size, hop, hp.max_time_steps, feat, hp.cin_pad, rank, world,
                                         train=False, n_classes=hp.quantize_channels)
    sync = D.GradSync(eng) if world > 1 else None

    max_steps = args.max_steps or hp.max_train_steps
    sched = getattr(lrschedule, hp.lr_schedule) if hp.lr_schedule else None
    t0, step0 = time.time(), step
    adam = adam_settings(hp)               # raises for anything the fused update is not (vqwae_train.py:1119-1120)
    lr = adam["lr"]
    try:
        while epoch < hp.nepochs and step < max_steps:
            run = torch.zeros(3, dtype=torch.float64, device=device)
            nb = 0
            for x, c, g, lengths in Prefetcher(loader, device):
                lr = adam["lr"]
                if sched is not None:
                    lr = sched(lr, step, **hp.lr_schedule_kwargs)                     # vqwae_train.py:729-735
                T = x.shape[1]
                ln = None if bool((lengths == T).all()) else lengths
                # ragged shards: the CE is normalised by the mask sum of the GLOBAL batch (vqwae_train.py:374-379 after :705)
                # (every rank enters the mask-sum all-reduce or none does: decided by the preset, not by this rank's shard)
                ce_scale, n_glob = D.step_ce_scale(lengths, T, x.shape[0], variable_length=hp.max_time_steps is None)
                res = eng.train_step(x, c, g, lengths=ln, lr=lr, betas=adam["betas"], eps=adam["eps"],
                                     weight_decay=adam["weight_decay"], clip_thresh=hp.clip_thresh,
                                     ema_decay=hp.ema_decay, grad_sync=sync, ce_scale=ce_scale,
                                     quantize_channels=hp.quantize_channels, log_scale_min=hp.log_scale_min)
                step += 1
                ce = res["ce"].float() * ce_scale               # rank mean of this = the global masked mean
                vq = res.get("vq_loss", torch.zeros((), device=device)).float()
                run += torch.stack([ce + vq, vq, res.get("perp", torch.zeros((), device=device)).float()]).double()
                nb += 1
                if step % 10 == 0 or step == 1:
                    eng.check_errors()          # ids the kernels had to clamp -> IndexError (the host reads scalars here anyway)
                    stats = torch.stack([ce + vq, vq, res.get("perp", torch.zeros((), device=device)).float()])
                    D.all_reduce_scalars(stats)
                    if rank == 0:
                        dt = (time.time() - t0) / max(step - step0, 1)
                        print(f"step {step} loss {float(stats[0]):.4f} vq {float(stats[1]):.4f} perp {float(stats[2]):.2f} "
                              f"gnorm {float(res['grad_norm']):.3f} lr {lr:.2e} {hp.batch_size * T / dt / 1e6:.2f} Msamples/s")
                if step > 0 and step % hp.train_eval_interval == 0 and rank == 0:                  # vqwae_train.py:834-836,772-774
                    print("[train_no_dev] Eval at train step {}".format(step))
                    eval_model(eng, x, c, g, lengths, step, os.path.join(args.checkpoint_dir, "intermediate", "train_no_dev_eval"),
                               hp, use_ema, hop)
                if step % hp.checkpoint_interval == 0:
                    eng.check_errors()          # never checkpoint weights that were trained on clamped ids
                    save_checkpoint(eng, step, epoch, args.checkpoint_dir, hp, rank, test_step, lr)
                if step >= max_steps:
                    print("Training reached max train steps ({}). will exit".format(max_steps)) if rank == 0 else None
                    break
            if nb:
                avg = (run / nb).float()
                D.all_reduce_scalars(avg)
                if rank == 0:                                                         # vqwae_train.py:862-869
                    print("Step {} [train_no_dev] Loss: {} vq: {} perp {}".format(step, float(avg[0]), float(avg[1]), float(avg[2])))
            if dev_loader is not None and step < max_steps:
                if epoch % hp.test_eval_epoch_interval == 0:                                          # :838-842: once per dev epoch
                    # EVERY rank draws this pass of the dev loader (its shuffle comes from one stateful generator that must stay
                    # identical on all ranks: evaluate() below slices the same permutation); rank 0 alone decodes
                    for xd, cd, gd, ld in Prefetcher(dev_loader, device):
                        if rank == 0:
                            print("[dev] Eval at train step {}".format(step))
                            eval_model(eng, xd, cd, gd, ld, step, os.path.join(args.checkpoint_dir, "intermediate", "dev_eval"), hp,
                                       use_ema, hop)
                        break
                dl, dvq, dperp = evaluate(eng, Prefetcher(dev_loader, device), device, hp)
                test_step += len(dev_loader)
                if rank == 0:
                    print("Step {} [dev] Loss: {} vq: {} perp {}".format(step, dl, dvq, dperp))
            epoch += 1
    except KeyboardInterrupt:
        print("Interrupted!")
    finally:
        save_checkpoint(eng, step, epoch, args.checkpoint_dir, hp, rank, test_step, lr)   # vqwae_train.py:1140-1145
    if rank == 0:
        print("Finished")
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python
"""Inference Rep from trained Autoencoder (entry point of the reference's inference_2019.py:2-11).

usage: inference_2019.py [options] <scp_dir> <feat> <checkpoint> <dst_dir>

    <scp_dir>     json list of [name, base_dir] pairs; base_dir ends in '/' and has six '/'-separated parts
                  (.../<lan>/<set>/<utterance>/): part -4 is the language, part -2 the utterance  (:226-230)
    <feat>        feature file stem inside base_dir ('mfcc.norm' -> <base_dir>mfcc.norm.npy, (N, 39) float32)
    <dst_dir>     output root; one text file <dst_dir>2019/<lan>/test/<utterance>.txt per utterance, one latent frame
                  per line, '%.6f' (:249-262)

options:
    --hparams=<parmas>       Hyper parameters [default: ].
    --preset=<json>          Path of preset parameters (json).

Encoder + quantizer only (vqvae_model.py:80-84), both on the HIP kernels; no CPU path.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from wavenet_autoencoders_amd.hparams import hparams  # noqa: E402


def output_path(base_dir, dst_dir):
    """inference_2019.py:226-230 (the reference concatenates strings, so dst_dir needs its trailing '/')."""
    dirs = base_dir.split("/")
    assert len(dirs) == 6, f"expected six '/'-separated parts in {base_dir!r}"
    return dst_dir + f"2019/{dirs[-4]}/test/{dirs[-2]}.txt"


def encode_features(eng, feat):
    """feat (N, c_in) float32 -> quantised latents (N', Cc) float32 (inference_2019.py:232-247)."""
    x = torch.from_numpy(np.ascontiguousarray(feat.T[None]).astype(np.float32)).to(eng.device)     # (1, c_in, N)
    if eng.weights_dirty:
        eng.prepare_weights()
    quant = eng.vq_forward(eng.encoder_forward(x))[0]
    return quant[0].t().contiguous().cpu().numpy()


def process_utterance(base_dir, f, eng, dst_dir):
    feat_path = base_dir + f + ".npy"
    if not os.path.exists(feat_path):
        raise FileNotFoundError(feat_path)
    out_path = output_path(base_dir, dst_dir)
    rep = encode_features(eng, np.load(feat_path))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    np.savetxt(out_path, rep, fmt="%.6f")
    print(f"{rep.shape}: {out_path}", flush=True)
    return out_path


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for a in ("scp_dir", "feat", "checkpoint", "dst_dir"):
        ap.add_argument(a)
    ap.add_argument("--hparams", default="")
    ap.add_argument("--preset")
    args = ap.parse_args(argv)
    if args.preset:
        with open(args.preset) as f:
            hparams.parse_json(f.read())
    hparams.parse(args.hparams)
    from vqwae_train import build_geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    eng = WaeEngine(build_geometry(hparams), dtype="fp32")
    eng.load_state_dict(torch.load(args.checkpoint, map_location="cpu")["state_dict"])
    print("Load checkpoint from {}".format(args.checkpoint), flush=True)
    os.makedirs(args.dst_dir, exist_ok=True)
    with open(args.scp_dir) as f:
        file_list = json.load(f)
    for _, base_dir in file_list:
        process_utterance(base_dir, args.feat, eng, args.dst_dir)
    return 0


if __name__ == "__main__":
    sys.exit(main())

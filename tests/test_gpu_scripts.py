"""The reference's three entry points end to end on the GPU with a tiny model: vqwae_train.py on a dump
in the reference's on-disk format -> checkpoint in the reference's key layout -> synthesis.py (autoregressive decode to wav)
and inference_2019.py (latent export as '%.6f' text), the latter checked against the oracle run on the saved weights."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import wae_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HP = ("layers=4,residual_channels=32,gate_channels=64,skip_out_channels=32,encoder_hid=32,cin_channels=16,gin_channels=8,"
      "n_speakers=5,batch_size=2,max_time_steps=2560,checkpoint_interval=1000")


def _run(args, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return p.stdout


def test_train_synthesis_and_feature_export_scripts(tmp_path):
    rng = np.random.default_rng(5)
    dump = tmp_path / "dump"
    lines = []
    for u in range(3):
        d = dump / "train_no_dev" / f"utt{u}"
        d.mkdir(parents=True)
        n = 40 + 4 * u
        np.save(d / "wave.npy", rng.integers(0, 256, n * 160).astype(np.int16))
        np.save(d / "mfcc.norm.npy", rng.standard_normal((n, 39)).astype(np.float32))
        lines.append(f"utt{u}|{n}|{u}|dummy")
    (dump / "train_no_dev" / "train.txt").write_text("\n".join(lines) + "\n")
    preset = os.path.join(ROOT, "hps", "vqwae.json")
    ck = tmp_path / "ck"
    out = _run([os.path.join(ROOT, "vqwae_train.py"), "--dump-root", str(dump), "--checkpoint-dir", str(ck), "--preset", preset,
                "--hparams", HP + ",train_eval_interval=2", "--max-steps", "3", "--dtype", "fp32"], str(tmp_path))
    assert "Finished" in out and "step 1 loss" in out
    # in-training online decoding with the averaged weights (vqwae_train.py:572-640,772-774): step 2 wrote both waveforms
    assert "Eval at train step 2" in out and "Using averaged model for evaluation" in out
    from scipy.io import wavfile as _wf
    for tag in ("predicted", "target"):
        sr_e, y_e = _wf.read(ck / "intermediate" / "train_no_dev_eval" / f"step000000002_{tag}.wav")
        assert sr_e == 16000 and y_e.shape == (2560,) and float(np.abs(y_e.astype(np.float64)).max()) > 0
    for f in ("checkpoint_step000000003.pth", "checkpoint_latest.pth", "checkpoint_step000000003_ema.pth", "hparams.json"):
        assert (ck / f).exists(), f
    sd = torch.load(ck / "checkpoint_latest.pth", map_location="cpu")["state_dict"]
    assert "wavenet.conv_layers.3.conv.weight_g" in sd and "vq.embedding.weight" in sd       # reference key layout

    # --- synthesis.py (reference synthesis.py:2-17 argument list; output <dst>2019/<lan>/test/<tar>_<fid>.wav, :521)
    short = dump / "test" / "S0_0007"
    short.mkdir(parents=True)
    np.save(short / "mfcc.norm.npy", rng.standard_normal((7, 39)).astype(np.float32))     # 7 frames: zero-padded to 8 (:482-486)
    (tmp_path / "syn.txt").write_text("test/S0_0007 V1\n")
    (tmp_path / "spk.json").write_text(json.dumps({"V1": 2}))
    _run([os.path.join(ROOT, "synthesis.py"), str(dump), str(ck / "checkpoint_latest.pth"), "wav/", str(tmp_path / "syn.txt"),
          str(tmp_path / "spk.json"), "english", "160", "25", "0", "--preset", preset, "--hparams", HP], str(tmp_path))
    from scipy.io import wavfile
    sr, y = wavfile.read(tmp_path / "wav" / "2019" / "english" / "test" / "V1_0007.wav")
    assert sr == 16000 and y.shape == (8 * 160,) and np.isfinite(y).all() and float(np.abs(y).max()) > 0

    # --- inference_2019.py (reference inference_2019.py:2-11; base_dir has six '/'-separated parts, :226-228)
    base = "db/x/english/test/utt9/"
    (tmp_path / base).mkdir(parents=True)
    feat = rng.standard_normal((50, 39)).astype(np.float32)
    np.save(tmp_path / base / "mfcc.norm.npy", feat)
    (tmp_path / "scp.json").write_text(json.dumps([["utt9", base]]))
    _run([os.path.join(ROOT, "inference_2019.py"), "scp.json", "mfcc.norm", str(ck / "checkpoint_latest.pth"), "out/", "--preset", preset,
          "--hparams", HP], str(tmp_path))
    got = np.loadtxt(tmp_path / "out" / "2019" / "english" / "test" / "utt9.txt")
    want = O.vqvae_encode(sd, torch.from_numpy(feat.T[None].copy()))[0].t().numpy()
    assert got.shape == want.shape == (13, 16)
    assert np.abs(got - want).max() < 2e-6                                               # '%.6f' rounding

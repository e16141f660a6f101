"""The engine's launch-path switches: ONE place, read ONCE per engine (WaeEngine.__init__ -> eng.opt).

Every entry selects between two implementations that are both kept correct and both tested -- the alternative is the yardstick of
the default (or an opt-in that has not earned the default).  Nothing else in the host code reads the
environment: the experiment switches of rounds 1-4 (request schedules, pacing, launch shapes that measured equal or slower;
profiles/EXPERIMENT_LOG.md) are gone from the product; what remains of them in the kernels is reachable through the C ABI's own
arguments (include/wae.h) from tools/.

    variable               default   alternative
    WAE_TN_STREAM          1         0: weight gradients as one 128 x 128 tile launch per layer (what fp32 always runs)
    WAE_TN_STATIC          1         0: the any-shape stream-K launch (csrc/gemm_tn_stream.hip) instead of the static-schedule one
    WAE_TN_SWAP            1         0: static weight-gradient launch: the conditioning and the out + skip member of a team keep their job
                                     kind for the whole launch (1: they change places halfway through the team's share)
    WAE_TN_STATIC_HEAD     1         0: head + first-conv weight gradients on the tile launches, not as a group of the static launch
    WAE_HEAD_SPLIT         1         0: the one-kernel head (16-bit engines run GEMM 0 as its own wae_gemm_tm launch by default)
    WAE_HEAD_WIDE          0         1: the separate-launch head of skip widths > 256, forced onto narrow models (tests)
    WAE_GLU_PAIR           inference 0 / 1: workgroup barrier on every second weight chunk of the layer kernel never / always
    WAE_DP_SPLIT           1         0: data parallel: the gradient arena handed to the all-reduce once, at the end of the sweep
    WAE_DP_WIRE            fp32      bf16: data parallel: the gradient all-reduce carries bf16 copies (half the bytes per link)
    WAE_AR_COOP            1         0: autoregressive decoding on the one-CU kernel even for <= 8 utterances
    WAE_AR_COOP_C          32        cooperating workgroups per utterance (1..32)
    WAE_BWD_FUSED          auto      residual(l) + gate(l-1) of the backward sweep as one launch (csrc/glu_bwd.hip): auto = 16-bit engines
                                     where the kernel has an instantiation; 0: always the two wae_gemm_tm launches; 1: fp32 too
    WAE_BWD_FOLD_DC        1         0: dc = sum_l Wc_l^T dz_l as its own K = L * 2Hp launch instead of riding in the fused backward launches
    WAE_SIDE               1         0: every launch of a train step on one stream (1: the step's independent side work -- the upsampling network
                                     and the backward weight packing beside the forward weight packing, the front end's backward beside
                                     the scatter of the weight gradients -- runs on side streams; engine.py: branch)
    WAE_CHAINS             auto      the backward sweep (and, beyond one round of workgroups, the gated stack) as two half-batch chains of launches
                                     on two streams (engine.py: chain_plan): auto = 16-bit engines whose layer launch has >= 200 workgroups
                                     (forward: > 256); 1: always one chain of full-batch launches; 2: two chains in both directions whenever
                                     the batch has two clips
"""
import os
from dataclasses import dataclass


@dataclass
class EngineOptions:
    tn_stream: bool = True
    tn_static: bool = True
    tn_static_head: bool = True
    tn_swap: bool = True
    head_split: bool = True
    head_wide: bool = False
    glu_pair: str = "inference"
    dp_split: bool = True
    dp_wire: str = "fp32"
    ar_coop: bool = True
    ar_coop_c: int = 32
    bwd_fused: str = "auto"
    bwd_fold_dc: bool = True
    chains: str = "auto"
    side: bool = True

    @staticmethod
    def from_env() -> "EngineOptions":
        e = os.environ.get
        pair = e("WAE_GLU_PAIR", "inference")
        if pair not in ("inference", "0", "1"):
            raise ValueError(f"WAE_GLU_PAIR={pair!r}: 'inference', '0' or '1'")
        fused = e("WAE_BWD_FUSED", "auto")
        if fused not in ("auto", "0", "1"):
            raise ValueError(f"WAE_BWD_FUSED={fused!r}: 'auto', '0' or '1'")
        wire = e("WAE_DP_WIRE", "fp32")
        if wire not in ("fp32", "bf16"):
            raise ValueError(f"WAE_DP_WIRE={wire!r}: 'fp32' or 'bf16'")
        chains = e("WAE_CHAINS", "auto")
        if chains not in ("auto", "1", "2"):
            raise ValueError(f"WAE_CHAINS={chains!r}: 'auto', '1' or '2'")
        return EngineOptions(chains=chains, side=e("WAE_SIDE", "1") != "0", dp_wire=wire, tn_stream=e("WAE_TN_STREAM", "1") != "0", tn_static=e("WAE_TN_STATIC", "1") != "0",
                             tn_static_head=e("WAE_TN_STATIC_HEAD", "1") != "0", tn_swap=e("WAE_TN_SWAP", "1") != "0", head_split=e("WAE_HEAD_SPLIT", "1") != "0",
                             head_wide=e("WAE_HEAD_WIDE", "0") == "1", glu_pair=pair, dp_split=e("WAE_DP_SPLIT", "1") != "0",
                             ar_coop=e("WAE_AR_COOP", "1") != "0", ar_coop_c=int(e("WAE_AR_COOP_C", "32")),
                             bwd_fused=fused, bwd_fold_dc=e("WAE_BWD_FOLD_DC", "1") != "0")


def two_chains(mode: str, is16: bool, B: int, T: int, backward: bool) -> bool:
    """The rule behind WaeEngine.chain_plan: does a sweep of layer launches over a (B, T) batch run as two half-batch chains?
    mode: EngineOptions.chains.  A layer launch has B * ceil(T / 256) workgroups on 256 CUs.
      BACKWARD: from 200 workgroups per launch (C2: 5.26 -> 5.19 ms per step; C5's two-launch sweep: 13.0 -> 10.9 ms);
      FORWARD: only beyond one round of the machine (C5: 320 workgroups, 10.5 -> 8.4 ms).  Within one round two chains gain nothing
      (C2 training 5.166-5.189 ms per step with, 5.148-5.159 without; inference 1.475-1.482 against 1.457-1.463 ms);
      hps/vqwae.json's 160-workgroup launches: nothing in either direction."""
    if mode == "1" or B < 2 or not is16:
        return False
    if mode == "2":
        return True
    tiles = B * ((T + 255) // 256)
    return tiles >= 200 and (backward or tiles > 256)

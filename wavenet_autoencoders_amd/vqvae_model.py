"""Drop-in ``VQVAE`` = Encoder + VectorQuantize + WaveNet (reference: vqvae_model.py:27-84) on the MI355X engine.

``state_dict`` keys: ``encoder.net.<i>.conv.{weight,bias}``, ``encoder.lin.{weight,bias}``, ``vq.embedding.weight``,
``wavenet.*`` -- the reference's, so its checkpoints load unchanged."""
import torch

from . import packing as P
from .wavenet_vocoder._base import ArenaModel
from .wavenet_vocoder.wavenet import WaveNet, _ids_from_input, softmax_bct


class _VQVAEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, ids, c, gid, train, *params):
        eng = model.engine()
        out = eng.forward(ids, c, gid, want_logits=True, train=train, dropout_on=model.training)
        ctx.model, ctx.ids, ctx.gid = model, ids, gid
        ctx.gen, ctx.train = getattr(eng, "fwd_gen", 0), train
        eng.check_errors()     # IndexError for an id outside its table, like the reference's nn.Embedding (wavenet.py:185-187)
        return out["logits"], out["vq_loss"].reshape(()), out["perp"].reshape(())

    @staticmethod
    def backward(ctx, dy, dvq, dperp):
        from . import _lib as L
        from . import backward as BW
        model = ctx.model
        eng = model._engine
        if ctx.train and getattr(eng, "fwd_gen", 0) != ctx.gen:
            raise RuntimeError("backward through a forward whose saved activations were overwritten by a later training-mode forward "
                               "of the same model: call backward before the next forward")
        g = eng.g
        B, O, T = dy.shape
        ext = torch.zeros(B, T, g.Op, dtype=eng.tdtype, device=dy.device)
        dyc = dy.contiguous().float()
        if eng.grad_scale != 1.0:               # fp16 stack: its backward runs on loss-scaled gradients (engine.py: grad_scale)
            dyc = dyc * eng.grad_scale
        L.check(eng.lib.wae_to_btc(L.ptr(dyc), L.ptr(ext), B, O, T, g.Op, eng.dt, eng.stream()), "to_btc")
        dc = BW.decoder_backward(eng, ctx.ids, None, None, ctx.gid, None, ext_dy=ext)
        BW.frontend_backward(eng, dc, float(dvq) if dvq is not None else 0.0)
        BW.finish_grads(eng)
        _, views = model._grad_views(eng)
        return (None, None, None, None, None) + tuple(v.clone() for v in views)


class VQVAE(ArenaModel):
    """VQVAE(c_in, hid, K, wavenet, encoder_hid).forward(x, c, g, softmax=False) -> (y_hat, vq_loss, perp)."""

    def __init__(self, c_in=39, hid=64, K=256, wavenet=None, encoder_hid=768):
        super().__init__()
        if not isinstance(wavenet, WaveNet):
            raise TypeError("wavenet must be a wavenet_autoencoders_amd.wavenet_vocoder.WaveNet")
        wg = wavenet.geom
        if wg.Cc != hid:
            raise ValueError(f"wavenet.cin_channels ({wg.Cc}) must equal hid ({hid})")
        geom = P.Geometry(layers=wg.layers, stacks=wg.stacks, R=wg.R, G=wg.G, S=wg.S, O=wg.O, Cc=wg.Cc, Cg=wg.Cg, k=wg.k,
                          n_speakers=wg.n_speakers, upsample_scales=wg.upsample_scales, cin_pad=wg.cin_pad,
                          scalar_input=wg.scalar_input, use_speaker_embedding=wg.use_speaker_embedding, c_in=c_in,
                          encoder_hid=encoder_hid, K=K, conv_in=wg.conv_in, up_act=wg.up_act, up_act_slope=wg.up_act_slope)
        self.out_channels, self.scalar_input = wavenet.out_channels, wavenet.scalar_input
        self.dropout = float(getattr(wavenet, "dropout", 0.0))          # the decoder layers' dropout (modules.py:127-128)
        self._init_arena(geom, "")
        # keep the decoder's initial values (the reference builds the WaveNet first: vqwae_train.py:926-946)
        own = dict(self.named_parameters())
        for n, p in wavenet.named_parameters():
            own["wavenet." + n].data.copy_(p.data)

    def _params(self):
        params = dict(self.named_parameters())
        return [params[r] for r in self._pnames]

    def forward(self, x, c, g, softmax=False):
        ids = _ids_from_input(x, self.out_channels, self.scalar_input, self.engine())
        gid = g.reshape(-1) if g is not None else None
        params = self._params()
        train = torch.is_grad_enabled() and any(p.requires_grad for p in params)   # (inside Function.forward grad mode is off)
        train = train or (self.training and self.dropout > 0)                       # F.dropout follows module.training
        y, vq_loss, perp = _VQVAEFn.apply(self, ids, c.float(), gid, train, *params)
        if softmax:
            y = softmax_bct(y)
        return y, vq_loss, perp

    def incremental_forward(self, initial_input, c, g, T, softmax, quantize, tqdm, log_scale_min):
        """encoder -> VQ -> autoregressive decoder (vqvae_model.py:73-79)."""
        eng = self.engine()
        with torch.no_grad():
            if eng.weights_dirty:
                eng.prepare_weights()
            lat = eng.encoder_forward(c.float())
            quant, idx, stats = eng.vq_forward(lat)
            from .wavenet_vocoder.wavenet import _start_classes
            init = 127 if self.scalar_input else _start_classes(initial_input, self.out_channels, eng)     # one start class per utterance
            gid = g.reshape(-1) if g is not None else None
            if self.scalar_input:
                M = self.out_channels // 3
                dev = c.device
                out = eng.incremental_forward(quant, gid, int(T), mode="sample", log_scale_min=log_scale_min,
                                              u_mix=torch.rand(c.shape[0], int(T), M, device=dev) * (1 - 2e-5) + 1e-5,
                                              u_log=torch.rand(c.shape[0], int(T), device=dev) * (1 - 2e-5) + 1e-5)
                return out["x"].unsqueeze(1)
            if quantize:
                if not softmax:
                    raise ValueError("quantize=True draws from the softmax probabilities: pass softmax=True")
                out = eng.incremental_forward(quant, gid, int(T), mode="sample", init_idx=init)
                idxs = out["idx"].long()
                return torch.nn.functional.one_hot(idxs, self.out_channels).float().transpose(1, 2).contiguous()
            # quantize=False: the probability / logit rows are the outputs and the fed-back inputs (wavenet.py:303-305,335-338)
            return eng.incremental_forward(quant, gid, int(T), mode="probs" if softmax else "raw", init_idx=init)["logits"]

    def encode(self, x):
        """quantised latents of MFCC features (vqvae_model.py:80-84; inference_2019.py:243-262)."""
        eng = self.engine()
        with torch.no_grad():
            if eng.weights_dirty:
                eng.prepare_weights()
            lat = eng.encoder_forward(x.float())
            return eng.vq_forward(lat)[0]

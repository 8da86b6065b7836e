"""MFM hierarchical multi-modal fusion of the teacher (reference: teacher/code/model.py —
TrainablePositionalEncoding :1135-1151, ThreeTransforTemproal :1300-1331, TwoTransforFusion :1361-1392,
ThreeTRXShiftLoopTime.extract_feature :1648-1664).  Inference (eval mode, no grad), which is how the student path
consumes it: rgb / depth / flow ResNet-50 features [N,8,2048] -> fused teacher feature [N,8,2048].

Parameters live in the same torch containers as the reference (nn.Embedding, nn.LayerNorm,
nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model, nhead, batch_first=True)), nn.Linear) so the state_dict keys
(`three_fusion.transformer_encoder.layers.0.self_attn.in_proj_weight`, `fusion.f1.weight`, ...) match the reference's
checkpoints; the arithmetic runs on the HIP GEMM + LayerNorm + 8-token attention kernels."""
import ctypes

import torch
import torch.nn as nn

from .. import ops
from .._lib import lib


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def layernorm(x, gamma, beta, res=None, res_rows=0, out=None, ldy=None, col_off=0):
    rows, D = x.shape
    if out is None:
        out = torch.empty_like(x)
        ldy = D
    ops._chk(x, gamma, beta, res)
    y_ptr = ctypes.c_void_p(out.data_ptr() + 4 * col_off)
    lib().call("lmkd_layernorm_fwd", _p(x), _p(res), res_rows, _p(gamma), _p(beta), y_ptr, ldy, rows, D, ctypes.c_float(1e-5), _s())
    return out


class TrainablePositionalEncoding(nn.Module):
    """model.py:1135-1151: LayerNorm(x + Embedding(position)); dropout is identity at inference."""

    def __init__(self, max_position_embeddings, hidden_size, dropout=0.1):
        super().__init__()
        self.position_embeddings = nn.Embedding(max_position_embeddings, hidden_size)
        self.LayerNorm = nn.LayerNorm(hidden_size)
        self.dropout = nn.Dropout(dropout)

    def into(self, x2d, L, out, ld, col_off):
        """x2d [N*L, D] -> out[:, col_off:col_off+D] (row stride ld)"""
        emb = self.position_embeddings.weight[:L].contiguous()
        return layernorm(x2d, self.LayerNorm.weight, self.LayerNorm.bias, emb, L, out, ld, col_off)


def _encoder_layer(x, layer, L, nhead):
    """nn.TransformerEncoderLayer defaults: post-norm, ReLU, eps 1e-5, batch_first.  x [B*L, D]"""
    M, D = x.shape
    at = layer.self_attn
    # the four Linear layers: ops.linear_infer - the bf16-plane convolution kernels (1x1 convolution + bias / ReLU epilogue) in the
    # library's default arithmetic, the fp32 MFMA GEMM in the native mode
    qkv = ops.linear_infer(x, at.in_proj_weight, at.in_proj_bias)
    ctx = torch.empty_like(x)
    lib().call("lmkd_mha_small", _p(qkv), _p(ctx), M // L, L, D, nhead, _s())
    o = ops.linear_infer(ctx, at.out_proj.weight, at.out_proj.bias)
    x = layernorm(x, layer.norm1.weight, layer.norm1.bias, o, M)
    h = ops.linear_infer(x, layer.linear1.weight, layer.linear1.bias, relu=True)
    f = ops.linear_infer(h, layer.linear2.weight, layer.linear2.bias)
    return layernorm(x, layer.norm2.weight, layer.norm2.bias, f, M)


class _Fusion(nn.Module):
    def __init__(self, args, n_mod, in_channels, dropout):
        super().__init__()
        self.n_mod, self.in_channels = n_mod, in_channels
        for i in range(n_mod):
            setattr(self, "positionEncoding%d" % (i + 1), TrainablePositionalEncoding(args.seq_len, in_channels))
        encoder_layer = nn.TransformerEncoderLayer(d_model=in_channels * n_mod, nhead=n_mod, batch_first=True)
        self.transformer_encoder = nn.TransformerEncoder(encoder_layer, num_layers=args.trans_num)
        self.f1 = nn.Linear(in_channels * n_mod, in_channels)
        self.dropout = nn.Dropout(dropout)

    @torch.no_grad()
    def extract_feature(self, *mods):
        if self.training:
            raise NotImplementedError("the MFM fusion is on the hot path as a frozen feature extractor: call .eval()")
        N, L, d = mods[0].shape
        D = d * self.n_mod
        cat = torch.empty((N * L, D), dtype=torch.float32, device=mods[0].device)
        for i, m in enumerate(mods):
            getattr(self, "positionEncoding%d" % (i + 1)).into(m.reshape(N * L, d).contiguous(), L, cat, D, i * d)
        h = cat
        for layer in self.transformer_encoder.layers:
            h = _encoder_layer(h, layer, L, self.n_mod)
        return ops.linear_infer(h, self.f1.weight, self.f1.bias).reshape(N, L, d)


class ThreeTransforTemproal(_Fusion):
    """model.py:1300-1331 (d_model 6144, 3 heads)."""

    def __init__(self, args, out_channels=None, dropout=0.1, in_channels=2048):
        super().__init__(args, 3, in_channels, dropout)


class TwoTransforFusion(_Fusion):
    """model.py:1361-1392 (d_model 4096, 2 heads)."""

    def __init__(self, args, out_channels=None, dropout=0.1, in_channels=2048):
        super().__init__(args, 2, in_channels, dropout)


class ThreeTRXShiftLoopTime(nn.Module):
    """model.py:1588-1664, the parts the student path uses: `three_fusion`, `fusion`, `extract_feature`.
    (`bracnch`, the teacher's own TRX head, is the TRX_2fcsup_fixed classifier of model_select.)"""

    def __init__(self, args, in_channels=2048):
        super().__init__()
        self.args = args
        self.fusion = TwoTransforFusion(args, in_channels=in_channels)
        self.three_fusion = ThreeTransforTemproal(args, in_channels=in_channels)

    @torch.no_grad()
    def extract_feature(self, feature):
        L, d = self.args.seq_len, self.three_fusion.in_channels
        rgb = feature["rgb"].reshape(-1, L, d)
        depth = feature["depth"].reshape(-1, L, d)
        flow = feature["flow"].reshape(-1, L, d)
        s = self.args.shirt_num
        out = self.three_fusion.extract_feature(rgb, depth, flow)
        f2 = self.fusion.extract_feature(rgb, torch.cat((depth[:, s:], depth[:, :s]), dim=1).contiguous())
        f3 = self.fusion.extract_feature(rgb, torch.cat((flow[:, s:], flow[:, :s]), dim=1).contiguous())
        for f in (f2, f3):
            lib().call("lmkd_axpby", _p(f), _p(out), ctypes.c_float(1.0), ctypes.c_float(1.0), out.numel(), _s())
        return out

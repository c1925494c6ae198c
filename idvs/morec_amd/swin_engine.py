"""Host-side orchestration of the vision item tower: explicit forward and hand-derived backward of
``Vit_Encoder.forward`` (``V/model/encoders.py:24-31``) = ``GELU(SwinForImageClassification(x)[0])``
(HF ``transformers/models/swin/modeling_swin.py``; the model is built at ``V/run.py:47-54``).

Layout: the residual stream is ``[n_img * H * W, C]`` rows in natural (image, y, x) order for the whole stage; window
partition / cyclic shift / window reverse never move data (the attention kernel computes the row of every window
token).  Pre-LN blocks: every residual sum is produced by the LayerNorm kernel that consumes it
(``z = res + droppath * (x + bias)``, ``y = LN(z)``), so a block costs two LayerNorm launches, four GEMMs and one
attention launch.  DropPath (stochastic depth, attention branch only -- ``modeling_swin.py:560``) is a per-image scale
vector from the counter-based RNG; the backward pass re-reads the same vector.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import ops
from ._lib import ACT_GELU, DACT_MUL
from .engine import NO_DROP, DropCfg, linear_wgrad_, prepare_linear

IN = "cv_encoder.image_net."


@dataclass
class SwinShape:
    image_size: int = 224
    patch_size: int = 4
    num_channels: int = 3
    embed_dim: int = 96
    depths: tuple = (2, 2, 6, 2)
    num_heads: tuple = (3, 6, 12, 24)
    window_size: int = 7
    mlp_ratio: float = 4.0
    layer_norm_eps: float = 1e-5
    drop_path_rate: float = 0.1

    @staticmethod
    def named(name: str) -> "SwinShape":
        """Shapes selected the way ``V/run.py:47-49`` keys on ``CV_model_load`` (pretrained_models/<name>/config.json)."""
        table = {
            "swin_tiny": dict(),
            "swin_small": dict(depths=(2, 2, 18, 2)),
            "swin_base": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32)),
            "swin_micro": dict(image_size=56, embed_dim=32, depths=(2, 2), num_heads=(1, 2)),
        }
        for key, kw in table.items():
            if key in name:
                return SwinShape(**kw)
        raise ValueError(f"unknown CV_model_load {name!r}")

    def drop_path_rates(self):
        n = sum(self.depths)
        return [self.drop_path_rate * i / max(n - 1, 1) for i in range(n)]     # modeling_swin.py:758

    def stage_geometry(self):
        """[(C, H, W, window, heads)] per stage."""
        g = self.image_size // self.patch_size
        out = []
        for s in range(len(self.depths)):
            out.append((self.embed_dim * 2 ** s, g >> s, g >> s, self.window_size, self.num_heads[s]))
        return out


def swin_layer_names(prefix: str, s: int, b: int) -> str:
    return prefix + f"swin.encoder.layers.{s}.blocks.{b}."


def swin_param_shapes(shape: SwinShape, D: int, prefix: str = IN):
    """``state_dict`` inventory of ``SwinForImageClassification`` with the replaced classifier (``V/run.py:50-51``), in
    the installed-HF registration order (``V/run.py:58-60`` freezes by parameter index)."""
    from collections import OrderedDict
    sw = prefix + "swin."
    out = OrderedDict()
    out[sw + "embeddings.patch_embeddings.projection.weight"] = (shape.embed_dim, shape.num_channels, shape.patch_size, shape.patch_size)
    out[sw + "embeddings.patch_embeddings.projection.bias"] = (shape.embed_dim,)
    out[sw + "embeddings.norm.weight"] = (shape.embed_dim,)
    out[sw + "embeddings.norm.bias"] = (shape.embed_dim,)
    C = shape.embed_dim
    for s, depth in enumerate(shape.depths):
        for b in range(depth):
            L = swin_layer_names(prefix, s, b)
            for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
                out[L + f"attention.{n}.weight"], out[L + f"attention.{n}.bias"] = (C, C), (C,)
            out[L + "attention.relative_position_bias.relative_position_bias_table"] = ((2 * shape.window_size - 1) ** 2, shape.num_heads[s])
            for n in ("layernorm_before", "layernorm_after"):
                out[L + n + ".weight"], out[L + n + ".bias"] = (C,), (C,)
            I = int(shape.mlp_ratio * C)
            out[L + "mlp.fc1.weight"], out[L + "mlp.fc1.bias"] = (I, C), (I,)
            out[L + "mlp.fc2.weight"], out[L + "mlp.fc2.bias"] = (C, I), (C,)
        if s < len(shape.depths) - 1:
            Dn = sw + f"encoder.layers.{s}.downsample."
            out[Dn + "reduction.weight"], out[Dn + "norm.weight"], out[Dn + "norm.bias"] = (2 * C, 4 * C), (4 * C,), (4 * C,)
            C *= 2
    out[sw + "layernorm.weight"], out[sw + "layernorm.bias"] = (C,), (C,)
    out[prefix + "classifier.weight"], out[prefix + "classifier.bias"] = (D, C), (D,)
    return out


# ---------------------------------------------------------------------------------------------------------
def swin_prepare(p: dict, shape: SwinShape, dtype, prefix: str = IN, shadow: dict | None = None):
    sh = shadow or {}
    sw = prefix + "swin."
    pw = p[sw + "embeddings.patch_embeddings.projection.weight"]
    prep = dict(patch=prepare_linear(pw.reshape(pw.shape[0], -1), dtype), stages=[], merges=[])
    for s, depth in enumerate(shape.depths):
        blocks = []
        for b in range(depth):
            L = swin_layer_names(prefix, s, b)
            A = L + "attention."
            wqkv, bqkv = p.get(A + "qkv_fused.weight"), p.get(A + "qkv_fused.bias")
            if wqkv is None:
                wqkv = torch.cat([p[A + f"{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")], 0)
                bqkv = torch.cat([p[A + f"{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")], 0)
            blocks.append(dict(qkv=prepare_linear(wqkv, dtype, sh.get(A + "qkv_fused.weight"), sh.get(A + "qkv_fused.weight" + "^T")), bqkv=bqkv,
                               o=prepare_linear(p[A + "o_proj.weight"], dtype, sh.get(A + "o_proj.weight"), sh.get(A + "o_proj.weight" + "^T")),
                               f1=prepare_linear(p[L + "mlp.fc1.weight"], dtype, sh.get(L + "mlp.fc1.weight"), sh.get(L + "mlp.fc1.weight" + "^T")),
                               f2=prepare_linear(p[L + "mlp.fc2.weight"], dtype, sh.get(L + "mlp.fc2.weight"), sh.get(L + "mlp.fc2.weight" + "^T"))))
        prep["stages"].append(blocks)
        if s < len(shape.depths) - 1:
            k = sw + f"encoder.layers.{s}.downsample.reduction.weight"
            prep["merges"].append(prepare_linear(p[k], dtype, sh.get(k), sh.get(k + "^T")))
    prep["cls"] = prepare_linear(p[prefix + "classifier.weight"], dtype, sh.get(prefix + "classifier.weight"), sh.get(prefix + "classifier.weight" + "^T"))
    return prep


def _window_of(H, W, ws, shift):
    if min(H, W) <= ws:      # modeling_swin.py:574-581: the window covers the whole map, no shift
        return min(H, W), 0
    return ws, shift


def swin_forward(p: dict, prep, shape: SwinShape, pixels: torch.Tensor, dtype, need_grad: bool, prefix: str = IN,
                 drop: DropCfg = NO_DROP, training: bool = False):
    """pixels fp32 [n_img, 3, R, R] (normalised, as V/run.py:201-204 uploads) or uint8 [n_img, R, R, 3] (decoded) -> item vectors [n_img, D] (``GELU(classifier(pool(LN(encoder(embed(x))))))``).
    ``training``: apply DropPath with the rates of ``shape.drop_path_rates()`` and the streams of ``drop``."""
    sw = prefix + "swin."
    n_img = pixels.shape[0]
    eps = shape.layer_norm_eps
    if hasattr(pixels, "patches"):    # data_utils.images.PatchRows: the normalising im2col already ran (on the input feed's stream, under the previous step)
        patches = pixels.patches
        assert patches.dtype == dtype and patches.shape[0] == n_img * (shape.image_size // shape.patch_size) ** 2
    elif pixels.dtype == torch.uint8:   # decoded HWC images: ToTensor + Normalize(0.5, 0.5) fused into the im2col (§8(f)-3)
        patches = ops.swin_patchify_u8(pixels.contiguous(), shape.patch_size, dtype)
    else:
        patches = ops.swin_patchify(pixels.contiguous(), shape.patch_size, dtype)
    e = ops.gemm_nt(patches, prep["patch"].w, bias=p[sw + "embeddings.patch_embeddings.projection.bias"])
    x, _, mean_e, rstd_e = ops.layernorm_fwd(e, p[sw + "embeddings.norm.weight"], p[sw + "embeddings.norm.bias"], 1e-5,
                                             save_z=False)
    saved_embed = (patches, e, mean_e, rstd_e)
    rates = shape.drop_path_rates()
    geom = shape.stage_geometry()
    pending = None          # (f, b2, h): the previous block's MLP output whose residual sum the next LayerNorm forms
    saved_stages, saved_merges = [], []
    li = 0
    for s, depth in enumerate(shape.depths):
        C, H, W, ws0, heads = geom[s]
        tokens = H * W
        saved_blocks = []
        for b in range(depth):
            L = swin_layer_names(prefix, s, b)
            w = prep["stages"][s][b]
            ws, shift = _window_of(H, W, ws0, 0 if b % 2 == 0 else ws0 // 2)
            g1, b1 = p[L + "layernorm_before.weight"], p[L + "layernorm_before.bias"]
            if pending is None:
                xn, _, mean1, rstd1 = ops.layernorm_fwd(x, g1, b1, eps, save_z=False)
            else:
                f, b2, h = pending
                xn, x, mean1, rstd1 = ops.layernorm_fwd(f, g1, b1, eps, bias=b2, res=h, z_inplace=True)
            desc = ops.swin_attn_desc(n_img, H, W, ws, shift, heads, C // heads, dtype)
            bias_t = ops.swin_bias_expand(p[L + "attention.relative_position_bias.relative_position_bias_table"], ws)
            qkv = ops.gemm_nt(xn, w["qkv"].w, bias=w["bqkv"])
            ctx = ops.swin_attn_fwd(desc, qkv, bias_t)
            a = ops.gemm_nt(ctx, w["o"].w)
            rate = rates[li] if training else 0.0
            scale = ops.droppath_scale(n_img, rate, drop.site(li), pixels.device) if rate > 0 else None
            hn, h, mean2, rstd2 = ops.layernorm_fwd(a, p[L + "layernorm_after.weight"], p[L + "layernorm_after.bias"], eps,
                                                    bias=p[L + "attention.o_proj.bias"], res=x, z_inplace=True,
                                                    rowscale=scale, rows_per_scale=tokens)
            # stage-1 / stage-2 widths (C <= 192): no act'(pre) tensor -- the backward recomputes the pre-activation from hn (ops.mlp_dact_recompute)
            b1 = p[L + "mlp.fc1.bias"]
            keep_pre = need_grad and not (b1.data_ptr() % 16 == 0 and ops.mlp_dact_recompute_supported(x.shape[0], w["f1"].w.shape[0], C, dtype))
            pre = torch.empty((x.shape[0], w["f1"].w.shape[0]), device=x.device, dtype=dtype) if keep_pre else None
            g = ops.gemm_nt(hn, w["f1"].w, bias=b1, act=ACT_GELU, aux_out=pre, aux_deriv=keep_pre)   # pre = GELU'(fc1)
            f = ops.gemm_nt(g, w["f2"].w)
            pending = (f, p[L + "mlp.fc2.bias"], h)
            saved_blocks.append((desc, bias_t, x, xn, mean1, rstd1, qkv, ctx, h, hn, mean2, rstd2, pre, g, scale, tokens))
            li += 1
        saved_stages.append(saved_blocks)
        if s < len(shape.depths) - 1:
            Dn = sw + f"encoder.layers.{s}.downsample."
            f, b2, h = pending
            out = ops.bias_residual(f, b2, h)
            m = ops.swin_merge(out, n_img, H, W, C)
            mn, _, mean_m, rstd_m = ops.layernorm_fwd(m, p[Dn + "norm.weight"], p[Dn + "norm.bias"], 1e-5, save_z=False)
            x = ops.gemm_nt(mn, prep["merges"][s].w)
            saved_merges.append((m, mn, mean_m, rstd_m, H, W, C))
            pending = None
    C, H, W, _, _ = geom[-1]
    f, b2, h = pending
    xf, z_last, mean_f, rstd_f = ops.layernorm_fwd(f, p[sw + "layernorm.weight"], p[sw + "layernorm.bias"], eps, bias=b2, res=h,
                                                   z_inplace=True)
    pooled = ops.swin_pool_fwd(xf, n_img, H * W)
    D = prep["cls"].w.shape[0]
    pre_c = torch.empty((n_img, D), device=x.device, dtype=dtype) if need_grad else None
    item = ops.gemm_nt(pooled, prep["cls"].w, bias=p[prefix + "classifier.bias"], act=ACT_GELU, aux_out=pre_c)
    saved = (shape, n_img, saved_embed, saved_stages, saved_merges, z_last, mean_f, rstd_f, pooled, pre_c) if need_grad else None
    return item, saved


def swin_backward(p: dict, prep, saved, d_item: torch.Tensor, grads: dict, prefix: str = IN, on_ready=None):
    """grads: name -> fp32 buffer (accumulated into); q/k/v projections either as ``attention.qkv_fused.{weight,bias}`` blocks
    (arena) or handed out as row views of a fresh fused buffer.  ``on_ready(("stage", s))`` (optional) is called once the
    norm / attention / downsample gradients of stage ``s`` are final (its ``mlp.fc2.bias`` entries are not: the first block's
    successor writes them), so a data-parallel driver can reduce them under the remaining stages."""
    sw = prefix + "swin."
    shape, n_img, saved_embed, saved_stages, saved_merges, z_last, mean_f, rstd_f, pooled, pre_c = saved
    geom = shape.stage_geometry()
    dv = ops.act_bwd(d_item.contiguous(), pre_c, ACT_GELU)
    ops.colsum_(dv, grads[prefix + "classifier.bias"])
    linear_wgrad_(dv, pooled, grads[prefix + "classifier.weight"])
    C, H, W, _, _ = geom[-1]
    dpooled = ops.gemm_nt(dv, prep["cls"].wt, K=dv.shape[1], N=C)
    dxf = ops.swin_pool_bwd(dpooled, n_img, H * W)
    last_s, last_b = len(shape.depths) - 1, shape.depths[-1] - 1
    b2_name = swin_layer_names(prefix, last_s, last_b) + "mlp.fc2.bias"
    # gradient at the last block's output sum (= the final LayerNorm's input); its column sums are that block's fc2.bias grad
    dout, _ = ops.layernorm_bwd(dxf, None, z_last, mean_f, rstd_f, p[sw + "layernorm.weight"], grads[sw + "layernorm.weight"],
                                grads[sw + "layernorm.bias"], dbias=grads[b2_name])
    for s in reversed(range(len(shape.depths))):
        C, H, W, _, heads = geom[s]
        if s < len(shape.depths) - 1:
            Dn = sw + f"encoder.layers.{s}.downsample."
            m, mn, mean_m, rstd_m, _, _, _ = saved_merges[s]
            linear_wgrad_(dout, mn, grads[Dn + "reduction.weight"])          # here dout = gradient of the next stage's input
            dmn = ops.gemm_nt(dout, prep["merges"][s].wt, K=dout.shape[1], N=4 * C)
            dm, _ = ops.layernorm_bwd(dmn, None, m, mean_m, rstd_m, p[Dn + "norm.weight"], grads[Dn + "norm.weight"],
                                      grads[Dn + "norm.bias"])
            dout = ops.swin_merge(dm, n_img, H, W, C, reverse=True)
            ops.colsum_(dout, grads[swin_layer_names(prefix, s, shape.depths[s] - 1) + "mlp.fc2.bias"])
        for b in reversed(range(shape.depths[s])):
            L = swin_layer_names(prefix, s, b)
            A = L + "attention."
            w = prep["stages"][s][b]
            desc, bias_t, x, xn, mean1, rstd1, qkv, ctx, h, hn, mean2, rstd2, pre, g, scale, tokens = saved_stages[s][b]
            # MLP branch: out = h + fc2(gelu(fc1(LN2(h)))) + b2
            linear_wgrad_(dout, g, grads[L + "mlp.fc2.weight"])
            if pre is None:
                du = ops.mlp_dact_recompute(dout, w["f2"].wt, hn, w["f1"].w, p[L + "mlp.fc1.bias"], colsum_out=grads[L + "mlp.fc1.bias"])
            else:
                du = ops.gemm_nt(dout, w["f2"].wt, dact=DACT_MUL, dact_in=pre, K=dout.shape[1], N=pre.shape[1],
                                 colsum_out=grads[L + "mlp.fc1.bias"])
            linear_wgrad_(du, hn, grads[L + "mlp.fc1.weight"])
            dhn = ops.gemm_nt(du, w["f1"].wt, K=du.shape[1], N=C)
            # h = x + droppath * (o_proj(ctx) + bo): dh = LN2'(dhn) + dout; da = droppath * dh
            dh, da = ops.layernorm_bwd(dhn, None, h, mean2, rstd2, p[L + "layernorm_after.weight"], grads[L + "layernorm_after.weight"],
                                       grads[L + "layernorm_after.bias"], dbias=grads[A + "o_proj.bias"], dres=dout,
                                       rowscale=scale, rows_per_scale=tokens)
            linear_wgrad_(da, ctx, grads[A + "o_proj.weight"])
            dctx = ops.gemm_nt(da, w["o"].wt, K=da.shape[1], N=C)
            dbias_t = torch.zeros_like(bias_t)
            gw, gb = grads.get(A + "qkv_fused.weight"), grads.get(A + "qkv_fused.bias")
            fused = gw is not None
            if not fused:
                gw = torch.zeros((3 * C, C), device=dctx.device, dtype=torch.float32)
                gb = torch.zeros(3 * C, device=dctx.device, dtype=torch.float32)
            dqkv = ops.swin_attn_bwd(desc, qkv, bias_t, ctx, dctx, dbias_t, dbqkv=gb)      # + d(q|k|v bias) = column sums of dqkv
            ops.swin_bias_reduce_(dbias_t, grads[A + "relative_position_bias.relative_position_bias_table"], desc.window)
            linear_wgrad_(dqkv, xn, gw)
            if not fused:
                for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
                    grads[A + f"{n}.weight"], grads[A + f"{n}.bias"] = gw[i * C:(i + 1) * C], gb[i * C:(i + 1) * C]
            dxn = ops.gemm_nt(dqkv, w["qkv"].wt, K=dqkv.shape[1], N=C)
            # x is the previous block's output sum (or the stage input): its bias gradient is the previous block's fc2.bias
            prev_b2 = grads[swin_layer_names(prefix, s, b - 1) + "mlp.fc2.bias"] if b > 0 else None
            dout, _ = ops.layernorm_bwd(dxn, None, x, mean1, rstd1, p[L + "layernorm_before.weight"],
                                        grads[L + "layernorm_before.weight"], grads[L + "layernorm_before.bias"], dbias=prev_b2,
                                        dres=dh)
        if on_ready is not None:
            on_ready(("stage", s))
    patches, e, mean_e, rstd_e = saved_embed
    de, _ = ops.layernorm_bwd(dout, None, e, mean_e, rstd_e, p[sw + "embeddings.norm.weight"], grads[sw + "embeddings.norm.weight"],
                              grads[sw + "embeddings.norm.bias"], dbias=grads[sw + "embeddings.patch_embeddings.projection.bias"])
    gpw = grads[sw + "embeddings.patch_embeddings.projection.weight"]
    linear_wgrad_(de, patches, gpw.view(gpw.shape[0], -1))

"""CPU restatement of FourCastNet / AFNONet (TEST INFRASTRUCTURE, see oracle/__init__.py).

State-dict driven restatement of reference models/fourcastnet/fourcastnet.py in eval mode:
`afno2d` = AFNO2D.forward (:78-127), `afno_block` = Block.forward (:180-193, double skip),
`afnonet_one_step` = patch embed (:530-543) + pos_embed + blocks (:283-293) + head + un-patchify
(:348-358).  PINNED by tests/golden/afno_*.npz (real reference, single-step calls driven from the
harness because the in-model multi-step loop crashes as shipped: `.to()` on a list, :336-340).

Quirks kept: the kept-mode count along W is derived from H (`total_modes = H//2+1`, :93-96);
ReLU and softshrink act on real and imaginary parts separately; `self.norm` exists in the state
dict but is never applied (:283-293); LayerNorm eps = 1e-6 (:245).
"""
import torch
import torch.nn.functional as F

from .common import rollout


def afno2d(x, w1, b1, w2, b2, num_blocks, sparsity_threshold=0.01, hard_thresholding_fraction=1.0):
    """x [B,H,W,C] fp32 -> same shape (already includes `+ bias` of :127)."""
    bias = x
    b, h, w, c = x.shape
    bs = c // num_blocks
    xf = torch.fft.rfft2(x.float(), dim=(1, 2), norm="ortho").reshape(b, h, w // 2 + 1, num_blocks, bs)
    total = h // 2 + 1
    kept = int(total * hard_thresholding_fraction)
    rs, ce = slice(total - kept, total + kept), kept
    xr, xi = xf[:, rs, :ce].real, xf[:, rs, :ce].imag
    ein = lambda a, m: torch.einsum("...bi,bio->...bo", a, m)
    o1r = F.relu(ein(xr, w1[0]) - ein(xi, w1[1]) + b1[0])
    o1i = F.relu(ein(xi, w1[0]) + ein(xr, w1[1]) + b1[1])
    o2r = ein(o1r, w2[0]) - ein(o1i, w2[1]) + b2[0]
    o2i = ein(o1i, w2[0]) + ein(o1r, w2[1]) + b2[1]
    outr = torch.zeros(xf.shape)
    outi = torch.zeros(xf.shape)
    outr[:, rs, :ce] = o2r
    outi[:, rs, :ce] = o2i
    z = F.softshrink(torch.stack([outr, outi], dim=-1), lambd=sparsity_threshold)
    z = torch.view_as_complex(z).reshape(b, h, w // 2 + 1, c)
    y = torch.fft.irfft2(z, s=(h, w), dim=(1, 2), norm="ortho")
    return y.type(x.dtype) + bias


def afno_block(x, sd, prefix, cfg):
    ln = lambda t, n: F.layer_norm(t, (t.shape[-1],), sd[f"{prefix}.{n}.weight"], sd[f"{prefix}.{n}.bias"], 1e-6)
    res = x
    x = afno2d(ln(x, "norm1"), sd[prefix + ".filter.w1"], sd[prefix + ".filter.b1"], sd[prefix + ".filter.w2"],
               sd[prefix + ".filter.b2"], cfg["num_blocks"], cfg.get("sparsity_threshold", 0.01),
               cfg.get("hard_thresholding_fraction", 1.0))
    x = x + res            # double_skip (:187-189)
    res = x
    x = ln(x, "norm2")
    x = F.linear(x, sd[prefix + ".mlp.fc1.weight"], sd[prefix + ".mlp.fc1.bias"])
    x = F.linear(F.gelu(x), sd[prefix + ".mlp.fc2.weight"], sd[prefix + ".mlp.fc2.bias"])
    return x + res


def afnonet_one_step(sd, cfg, x):
    p1, p2 = cfg["patch_size"]
    b = x.shape[0]
    hh, ww = cfg["img_height"] // p1, cfg["img_width"] // p2
    x = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=(p1, p2))
    x = x.flatten(2).transpose(1, 2)
    if cfg.get("use_pos_embed", True):
        x = x + sd["pos_embed"]
    x = x.reshape(b, hh, ww, cfg["embed_dim"])
    for i in range(cfg["depth"]):
        x = afno_block(x, sd, f"blocks.{i}", cfg)
    x = F.linear(x, sd["head.weight"])
    cout = x.shape[-1] // (p1 * p2)
    # "b h w (p1 p2 c_out) -> b c_out (h p1) (w p2)"
    x = x.view(b, hh, ww, p1, p2, cout).permute(0, 5, 1, 3, 2, 4).reshape(b, cout, hh * p1, ww * p2)
    return x


def afnonet_rollout(sd, cfg, constants, prescribed, prognostic):
    return rollout(lambda xt: afnonet_one_step(sd, cfg, xt), cfg["context_size"], constants, prescribed, prognostic)

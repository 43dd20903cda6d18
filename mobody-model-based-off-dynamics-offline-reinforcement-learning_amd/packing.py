"""Reference state_dict tensors <-> the packed weight blobs the HIP kernels stream.

Plumbing only (torch pad/transposes, run once per load/save): the layouts come from the C
ABI (`mobody_dyn_layout`, `mobody_mlp_layout`), so there is one source of truth.

  dynamics : `<layer>.weight [E,in,out]`, `<layer>.bias [E,1,out]`  (mobody_module.py:383-389)
             -> per layer W[E][Kp][Np] (zero padded; only the inference halves of the
                latent / reward heads), bias[E][Np]
  MLP      : `network.{0,2,4}.weight [out,in]`, `.bias [out]`        (mobody.py:35-48)
             -> per member W1[Kp1][256] b1 W2[256][256] b2 W3[256][Np3] b3  (W = weight.T)
"""
import torch

from . import _lib

MLP_KEYS = ("network.0", "network.2", "network.4")


def wide_pack(W):
    """[..., K, 256] row-major -> the kernels' K-interleaved-by-4 storage [..., K/4, 256, 4] (csrc/tile.h wide_idx)."""
    *lead, K, N = W.shape
    assert N == 256 and K % 4 == 0
    return W.reshape(*lead, K // 4, 4, N).transpose(-1, -2).contiguous()


def wide_unpack(flat, K):
    """Inverse of wide_pack for one matrix: flat [K*256] -> [K, 256]."""
    return flat.reshape(K // 4, 256, 4).transpose(-1, -2).reshape(K, 256)


def pack_dynamics(params, S, A, device):
    """params: dict name -> array/tensor in the reference layout. Returns a flat fp32 device blob."""
    L = _lib.dyn_layout(S, A)
    blob = torch.zeros(L.total_floats, dtype=torch.float32, device=device)
    for li, name in enumerate(_lib.DL_NAMES):
        lay = L.layer[li]
        W = torch.as_tensor(params[name + ".weight"], dtype=torch.float32).to(device)
        b = torch.as_tensor(params[name + ".bias"], dtype=torch.float32).to(device)
        E, K, _ = W.shape
        assert E == L.E and K == lay.in_dim, (name, tuple(W.shape), lay.in_dim)
        n = lay.out_dim                                    # leading `out_dim` columns = the half inference uses
        Wp = torch.zeros(E, lay.Kp, lay.Np, dtype=torch.float32, device=device)
        Wp[:, :K, :n] = W[:, :, :n]
        bp = torch.zeros(E, lay.Np, dtype=torch.float32, device=device)
        bp[:, :n] = b[:, 0, :n]
        if lay.Np == 256:                                   # 256-wide matrices use the interleaved storage
            Wp = wide_pack(Wp)
        blob[lay.w_off:lay.w_off + Wp.numel()] = Wp.reshape(-1)
        blob[lay.b_off:lay.b_off + bp.numel()] = bp.reshape(-1)
    return blob


def pack_mlp(member_params, in_dim, out_dim, device, prefixes=None):
    """member_params: list (one per member) of dicts with `network.{0,2,4}.{weight,bias}` keys
    (nn.Linear layout), or one dict plus `prefixes` (e.g. ['network1.', 'network2.'] for twin-Q)."""
    if prefixes is not None:
        member_params = [{k[len(p):]: v for k, v in member_params.items() if k.startswith(p)} for p in prefixes]
    M = len(member_params)
    L = _lib.mlp_layout(in_dim, out_dim, M)
    blob = torch.zeros(L.total_floats, dtype=torch.float32, device=device)
    for m, p in enumerate(member_params):
        base = m * L.member_floats
        g = lambda k: torch.as_tensor(p[k], dtype=torch.float32).to(device)
        W1, W2, W3 = g("network.0.weight"), g("network.2.weight"), g("network.4.weight")
        assert W1.shape == (256, in_dim) and W2.shape == (256, 256) and W3.shape == (out_dim, 256), \
            "packed MLPs are in->256->256->out (hidden_sizes must be 256)"
        w1 = torch.zeros(L.Kp1, 256, device=device); w1[:in_dim] = W1.t()
        w3 = torch.zeros(256, L.Np3, device=device); w3[:, :out_dim] = W3.t()
        b3 = torch.zeros(L.Np3, device=device); b3[:out_dim] = g("network.4.bias")
        for off, t in ((L.w1, wide_pack(w1)), (L.b1, g("network.0.bias")), (L.w2, wide_pack(W2.t().contiguous())),
                       (L.b2, g("network.2.bias")), (L.w3, w3), (L.b3, b3)):
            blob[base + off:base + off + t.numel()] = t.reshape(-1)
    return blob


def unpack_mlp(blob, in_dim, out_dim, members):
    """Inverse of pack_mlp: list of dicts in nn.Linear layout (clones)."""
    L = _lib.mlp_layout(in_dim, out_dim, members)
    out = []
    for m in range(members):
        base = m * L.member_floats
        v = lambda off, n: blob[base + off:base + off + n]
        out.append({
            "network.0.weight": wide_unpack(v(L.w1, L.Kp1 * 256), L.Kp1)[:in_dim].t().contiguous(),
            "network.0.bias": v(L.b1, 256).clone(),
            "network.2.weight": wide_unpack(v(L.w2, 65536), 256).t().contiguous(),
            "network.2.bias": v(L.b2, 256).clone(),
            "network.4.weight": v(L.w3, 256 * L.Np3).view(256, L.Np3)[:, :out_dim].t().contiguous(),
            "network.4.bias": v(L.b3, L.Np3)[:out_dim].clone(),
        })
    return out


# ---- dynamics pre-training blob (csrc/pretrain.hip, MobodyPretrainLayout) ------------------------------------------
PRETRAIN_NETS = (("enc", ("zs1", "zs2", "zs3")), ("tr", ("transition1", "transition2", "transition3")),
                 ("rw", ("reward_model1", "reward_model2", "reward_model3")))
ZA_NETS = (("off_za_src", "za_src"), ("off_za_trg", "za_trg"))


def pack_pretrain(params, S, A, device):
    """Reference state_dict tensors (`<layer>.weight [7,in,out]`, `.bias [7,1,out]`) -> the training blob."""
    L = _lib.pretrain_layout(S, A)
    g = lambda k: torch.as_tensor(params[k], dtype=torch.float32).to(device)
    blob = torch.zeros(L.total_floats, dtype=torch.float32, device=device)
    for nm, (l1, l2, l3) in PRETRAIN_NETS:
        ml = getattr(L, nm)
        members = []
        for e in range(7):                              # EnsembleLinear stores [in, out]; pack_mlp wants nn.Linear's [out, in]
            members.append({"network.0.weight": g(l1 + ".weight")[e].t(), "network.0.bias": g(l1 + ".bias")[e, 0],
                            "network.2.weight": g(l2 + ".weight")[e].t(), "network.2.bias": g(l2 + ".bias")[e, 0],
                            "network.4.weight": g(l3 + ".weight")[e].t(), "network.4.bias": g(l3 + ".bias")[e, 0]})
        off = getattr(L, "off_" + nm)
        blob[off:off + ml.total_floats] = pack_mlp(members, ml.in_dim, ml.out_dim, device)
    for off_name, pre in ZA_NETS:
        off = getattr(L, off_name)
        W1, b1, W2, b2 = g(pre + "1.weight"), g(pre + "1.bias"), g(pre + "2.weight"), g(pre + "2.bias")
        z = torch.zeros(7, L.za_member_floats, dtype=torch.float32, device=device)
        z[:, L.za_w1:L.za_w1 + L.za_in * 32] = W1.reshape(7, -1)
        z[:, L.za_b1:L.za_b1 + 32] = b1[:, 0]
        z[:, L.za_w2:L.za_w2 + 512] = W2[:, :, :16].reshape(7, -1)          # mu half only
        z[:, L.za_b2:L.za_b2 + 16] = b2[:, 0, :16]
        blob[off:off + z.numel()] = z.reshape(-1)
    return blob


def unpack_pretrain(blob, S, A, into=None):
    """Inverse of pack_pretrain.  Returns {name: tensor} in the reference layout; `into` (a state-dict-like mapping of
    full-size tensors) supplies the halves the blob does not hold (logvar half of za_*2) and is updated in place."""
    L = _lib.pretrain_layout(S, A)
    out = {}
    for nm, names in PRETRAIN_NETS:
        ml = getattr(L, nm)
        off = getattr(L, "off_" + nm)
        ms = unpack_mlp(blob[off:off + ml.total_floats], ml.in_dim, ml.out_dim, 7)
        for li, lname in zip((0, 2, 4), names):
            out[lname + ".weight"] = torch.stack([m[f"network.{li}.weight"].t() for m in ms]).contiguous()
            out[lname + ".bias"] = torch.stack([m[f"network.{li}.bias"] for m in ms]).unsqueeze(1).contiguous()
    for off_name, pre in ZA_NETS:
        off = getattr(L, off_name)
        z = blob[off:off + 7 * L.za_member_floats].view(7, L.za_member_floats)
        out[pre + "1.weight"] = z[:, L.za_w1:L.za_w1 + L.za_in * 32].reshape(7, L.za_in, 32).clone()
        out[pre + "1.bias"] = z[:, L.za_b1:L.za_b1 + 32].reshape(7, 1, 32).clone()
        W2mu = z[:, L.za_w2:L.za_w2 + 512].reshape(7, 32, 16)
        b2mu = z[:, L.za_b2:L.za_b2 + 16].reshape(7, 1, 16)
        if into is not None:
            W2, b2 = into[pre + "2.weight"].clone(), into[pre + "2.bias"].clone()
            W2[:, :, :16] = W2mu; b2[:, :, :16] = b2mu
            out[pre + "2.weight"], out[pre + "2.bias"] = W2, b2
        else:
            out[pre + "2.weight.mu"], out[pre + "2.bias.mu"] = W2mu.clone(), b2mu.clone()
    if into is not None:
        for k, v in out.items():
            into[k] = v
    return out

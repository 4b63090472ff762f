"""Shared by make_golden.py and the tests: seeded input recipes and tensor digests.

A *recipe* regenerates an input from a CPU seed (torch's CPU RNG stream is stable); the fixture
keeps its digest so drift is detected.  A *digest* stands in for a tensor too large to commit:
(sum, |.|-sum, squares-sum in fp64 + 32-element head/tail).  Small tensors are stored whole.
"""
import torch

FULL_LIMIT = 4096


def rnd(*shape, seed, scale=1.0, shift=0.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale + shift


def digest(t):
    d = t.detach().double().reshape(-1)
    return dict(shape=tuple(t.shape), sum=float(d.sum()), abs_sum=float(d.abs().sum()), sq_sum=float((d * d).sum()),
                head=t.detach().reshape(-1)[:32].clone(), tail=t.detach().reshape(-1)[-32:].clone())


def pack(t, full=False):
    """Store small tensors whole, large ones as digests."""
    if full or t.numel() <= FULL_LIMIT:
        return t.detach().clone()
    return digest(t)


def pack_dict(d, full_keys=()):
    return {k: pack(v, full=k in full_keys) for k, v in d.items()}


def check(actual, packed, rtol, atol=0.0, name=""):
    """Assert ``actual`` matches a packed tensor/digest; tolerances are relative to the max magnitude."""
    if torch.is_tensor(packed):
        scale = float(packed.abs().max()) or 1.0
        err = float((actual.detach().cpu().to(packed.dtype) - packed).abs().max())
        assert err <= rtol * scale + atol, f"{name}: max err {err:.3e} vs scale {scale:.3e}"
        return err / scale
    a = actual.detach().cpu()
    assert tuple(a.shape) == packed["shape"], f"{name}: shape {tuple(a.shape)} != {packed['shape']}"
    got = digest(a)
    n = a.numel()
    rtol = max(rtol, 1e-12)          # fp64 reduction order differs with thread count
    rms = (packed["sq_sum"] / n) ** 0.5 or 1.0
    # sums of n terms each within rtol*rms: allow sqrt(n)-ish growth generously (n * rtol * rms is the hard bound)
    assert abs(got["sum"] - packed["sum"]) <= rtol * rms * n ** 0.5 * 4 + atol * n, f"{name}: sum {got['sum']} vs {packed['sum']}"
    assert abs(got["abs_sum"] - packed["abs_sum"]) <= rtol * packed["abs_sum"] + atol * n, f"{name}: abs_sum"
    assert abs(got["sq_sum"] - packed["sq_sum"]) <= 2 * rtol * packed["sq_sum"] + atol * n, f"{name}: sq_sum"
    scale = max(float(packed["head"].abs().max()), float(packed["tail"].abs().max()), rms)
    for part in ("head", "tail"):
        err = float((got[part].to(packed[part].dtype) - packed[part]).abs().max())
        assert err <= rtol * scale * 4 + atol, f"{name}: {part} err {err:.3e} vs scale {scale:.3e}"
    return 0.0


# ---- comparing parameter states after Adam steps (the G9 fixture; tiny UNet: hid 32, mult (1, 2))
def noise_driven(key):
    """Parameters whose gradient is analytically ZERO in the tiny net: a per-channel constant added in front of a
    GroupNorm with one channel per group (32 channels, 32 groups) — conv1.bias and the time-bias projection of the
    32-channel blocks, the output bias of the last block (it feeds only out_conv's GroupNorm) — and the key third of project_in.bias (a constant added to every key shifts all logits of a query
    alike: the softmax does not see it).  Their numerical gradient is rounding noise (~1e-9), which Adam normalises into
    +-lr steps of arbitrary sign, in the reference as much as here: they cannot be compared (and have no effect on any output)."""
    return (("level_0." in key and (key.endswith("conv1.bias") or ".fc." in key)) or key.endswith("project_in.bias")
            or key in ("upsamples.level_0.1.conv2.bias", "upsamples.level_0.1.skip.bias"))


def check_state(actual, expected, tol, name, adam_slack=0.0):
    """Every tensor within ``tol`` (relative to its max magnitude) — except that, when ``adam_slack`` > 0 (comparisons
    against another implementation's Adam trajectory), up to 2 % of a tensor's elements may be off by at most
    ``adam_slack``: Adam turns a gradient element that sits at rounding-noise level in the first steps into +-lr moves."""
    for k, v in expected.items():
        if noise_driven(k):
            continue
        if not torch.is_tensor(v):                      # large tensors are committed as digests (sums + head / tail)
            check(actual[k], v, 5e-3 if adam_slack > 0 else tol, name=f"{name}.{k}")
            continue
        err = (actual[k].detach().double() - v.double()).abs()
        scale = float(v.abs().max()) or 1.0
        bad = err > tol * scale
        assert float(err.max()) <= max(tol * scale, adam_slack), f"{name}.{k}: max err {float(err.max()):.3e}"
        assert int(bad.sum()) <= 0.02 * bad.numel() * (adam_slack > 0), f"{name}.{k}: {int(bad.sum())}/{bad.numel()} elements beyond {tol:g}"




def g15_inputs(D=48, seed=151):
    """Synthetic feature clouds of fixture G15 (evaluation arithmetic): recreated from the seed on both sides instead of being committed."""
    g = torch.Generator().manual_seed(seed)
    mix = torch.randn(D, D, generator=g) / D ** 0.5
    gen = torch.randn(700, D, generator=g) @ mix + 0.3
    real = torch.randn(900, D, generator=g) @ (mix * 1.1) + torch.linspace(-0.2, 0.4, D)
    return gen, real

"""TEST INFRASTRUCTURE shared by the GPU tests."""


def same_kernel_results(a, b, what=""):
    """Two launches of the SAME generated program through different kernel instantiations (streaming / write-back stores, paired / 8-byte stores, 32- / 64-bit
    offsets, another launch geometry): bit-identical, or -- where the compiler contracted `x y + z w` as fma(x, y, z w) in one instantiation and as fma(z, w, x y) in
    the other -- equal to the last bit or two: 1e-14 of the largest magnitude, identical zeros, no entry unwritten."""
    import torch
    if torch.equal(a, b):
        return True
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert not torch.isnan(a).any() and not torch.isnan(b).any(), f"{what}: unwritten entries"
    scale = max(float(b.abs().max()), 1e-300)
    worst = float((a - b).abs().max())
    assert worst <= 1e-14 * scale, f"{what}: results of two instantiations of one program differ by {worst} (scale {scale})"
    assert torch.equal(a == 0, b == 0), f"{what}: structural zeros differ"
    return True

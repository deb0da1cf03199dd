"""Feature rows narrower than four floats padded to four for the autograd operators."""
import torch

# Below this many rows a call is bound by launch latency, not by how its lanes move the rows:
# the two extra copies would only add launches (Cora-sized graphs).
PAD_MIN_ROWS = 1 << 15


def pad4(x):
    """[n,h,f] -> the same rows zero-padded to four features when f < 4, else unchanged.  Since
    round 3 the kernels move every row of at least four floats as float4 lanes whatever its
    width or alignment (16-byte global accesses at dword-aligned addresses run at full rate on
    gfx950: csrc/common.hpp ld4u, tools/probe_unaligned.hip), so 4 x 41 or 1 x 47 need no padded
    copies any more; only rows of one to three floats would be left scalar lanes.  Zero features
    gather, accumulate and dot to zero, so the first f features are unchanged."""
    f = x.size(2)
    if f >= 4 or x.size(0) < PAD_MIN_ROWS:
        return x
    return torch.nn.functional.pad(x, (0, 4 - f))


def pad_to(x, width):
    """[m,h,f] zero-padded to exactly `width` features: the output gradient of a padded forward.
    The forward decided from ITS tensor's row count whether to pad; the gradient has the row
    count of the other side of a rectangular block, so it follows the saved tensor's width
    instead of deciding again (m < PAD_MIN_ROWS <= n would leave the two at different widths)."""
    f = x.size(2)
    return x if f == width else torch.nn.functional.pad(x, (0, width - f))


def unpad(x, f):
    """View of the first f features (the caller's width)."""
    return x if x.size(2) == f else x[..., :f]

"""Tile farm: split one frame into horizontal bands, one per rank, and gather the RGB8 band images
onto the presenting rank.  Pure plumbing over torch.distributed (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in CPU tests); no pixel is computed here.

Bands are contiguous rows from the TOP of the view; the RGB8 image is bottom-up
(scene.cuh:2013-2015), so band r -- rows [y0, y0+rows) -- occupies byte rows
[H-(y0+rows), H-y0) of the assembled frame, and each band image is itself bottom-up, which makes
the assembled frame a plain concatenation of the band images in reverse rank order.

The exchange is one gather per frame (0.78 MB per rank at 1080p / 8 GPUs: latency-, not
bandwidth-bound on xGMI).  `gather_bands(..., async_op=True)` returns a handle so the caller can
render frame i+1 while frame i's bands are still in flight (double-buffered band images).
With overlapping frames (ptrt_set_option "pipeline", the default) a frame into image A waits only for
what was on the scene's stream when the PREVIOUS render call was made: the collective that still
reads A runs on the communicator's stream, so `.wait()` on its handle -- a wait of the current
stream, not of the host -- belongs before the render call in between (bench.py, Farm.step), not just
before the one that overwrites A.
"""


def bands(height, world):
    """[(y0, rows)] per rank; the last rank takes the remainder."""
    base = height // world
    out = []
    for r in range(world):
        y0 = r * base
        out.append((y0, base if r < world - 1 else height - y0))
    return out


def strip_rows(height, world, rank):
    """Frame rows (top-down) of the context that owns every world-th 8-row strip from strip `rank`
    (ptrt_create_interleaved): what balances a frame whose cost is uneven over its height."""
    rows = []
    for t in range(rank, (height + 7) // 8, world):
        rows.extend(range(8 * t, min(8 * t + 8, height)))
    return rows


def strip_frame_index(height, world, rank):
    """For the bottom-up RGB8 image of that context: the frame's (bottom-up) byte row of each of its byte rows."""
    rows = strip_rows(height, world, rank)
    return [height - 1 - y for y in reversed(rows)]


def max_strip_rows(height, world):
    return max(len(strip_rows(height, world, r)) for r in range(world))


def gather_strips(dist, tile, frame, parts, index, rank, world, async_op=False):
    """Collective for the strip layout: every rank contributes its (max_rows, W, 3) image (its rows first, padding
    behind); rank 0 receives all into `parts` and scatters the valid rows into `frame` (index[r]: LongTensor of frame
    rows).  With async_op the scatter is the caller's (`scatter_strips`) after .wait()."""
    if world == 1:
        return None
    w = dist.gather(tile, parts if rank == 0 else None, dst=0, async_op=async_op)
    if async_op:
        return _Works([w])
    if rank == 0:
        scatter_strips(frame, parts, index)
    return None


def scatter_strips(frame, parts, index):
    for r, idx in enumerate(index):
        frame.index_copy_(0, idx, parts[r][:idx.numel()])


class StripAssembler:
    """Rank 0's side of the strip layout with ONE copy kernel per frame: the ranks' (padded) strip images are gathered into
    views of one (world * max_rows, W, 3) buffer, and one index_copy_ moves every row to its place in a frame that has one
    spare row (index `height`) for the padding rows."""

    def __init__(self, torch, height, width, world, device):
        self.height, self.rows = height, max_strip_rows(height, world)
        self.buf = torch.empty((world * self.rows, width, 3), dtype=torch.uint8, device=device)
        self.parts = [self.buf[r * self.rows:(r + 1) * self.rows] for r in range(world)]
        idx = []
        for r in range(world):
            fi = strip_frame_index(height, world, r)
            idx += fi + [height] * (self.rows - len(fi))
        self.index = torch.tensor(idx, dtype=torch.long, device=device)

    def scatter(self, frame_ext):
        """frame_ext: (height + 1, W, 3); rows [0, height) are the assembled frame afterwards."""
        frame_ext.index_copy_(0, self.index, self.buf)


def frame_views(frame, height, world):
    """Views into a (H, W, 3) uint8 frame, one per rank, where that rank's band image lands."""
    return [frame[height - (y0 + rows):height - y0] for (y0, rows) in bands(height, world)]


class _Works:
    def __init__(self, works):
        self.works = [w for w in works if w is not None]

    def wait(self):
        for w in self.works:
            w.wait()


def gather_bands(dist, tile, views, rank, world, height, async_op=False):
    """Collective: every rank contributes `tile` (rows, W, 3); rank 0 receives all into `views`.
    Equal bands use one gather; a remainder band falls back to point-to-point.
    Returns None, or with async_op=True an object with .wait()."""
    if world == 1:
        return None
    if height % world == 0:
        w = dist.gather(tile, views if rank == 0 else None, dst=0, async_op=async_op)
        return _Works([w]) if async_op else None
    if rank == 0:
        views[0].copy_(tile)
        reqs = [dist.irecv(views[r], src=r) for r in range(1, world)]
    else:
        reqs = [dist.isend(tile, dst=0)]
    if async_op:
        return _Works(reqs)
    for q in reqs:
        q.wait()
    return None


# ---- denoiser / bloom on the presenting rank -------------------------------------------------------
# The spatiotemporal denoiser reads +-32 rows around a pixel (a-trous step 16) and previous-frame history, bloom
# a six-level mip chain of the whole frame: both run where the whole frame is.  The bands therefore send their HDR
# image and G-buffers instead of RGB8 -- accum 12 + normal 12 + depth 4 + objectId 4 = 32 B/px, 8.3 MB per rank at
# 1080p / 8 GPUs, still ~60 us per link on xGMI -- and rank 0 runs `Scene.post_frame` (ptrt_post_frame) over them.
# HDR and G-buffers are top-down, so the gathered frame is the plain concatenation of the bands in rank order.
GBUFFER_KINDS = (("accum", 3, "float32"), ("normal", 3, "float32"), ("depth", 1, "float32"), ("object_id", 1, "int32"))


def gbuffer_views(frames, height, world):
    """frames: dict kind -> (H*W, c) tensor on rank 0; returns per kind the list of per-rank row-range views."""
    out = {}
    for kind, _, _ in GBUFFER_KINDS:
        t = frames[kind]
        w = t.shape[0] // height
        out[kind] = [t[y0 * w:(y0 + rows) * w] for (y0, rows) in bands(height, world)]
    return out


def gather_gbuffers(dist, band, views, rank, world, height, async_op=False):
    """Collective: every rank contributes its band's four buffers (dict kind -> (rows*W, c) tensor); rank 0
    receives them into `views` (from gbuffer_views).  Returns None or an object with .wait()."""
    if world == 1:
        return None
    works = []
    for kind, _, _ in GBUFFER_KINDS:
        if height % world == 0:
            works.append(dist.gather(band[kind], views[kind] if rank == 0 else None, dst=0, async_op=async_op))
        elif rank == 0:
            views[kind][0].copy_(band[kind])
            works += [dist.irecv(views[kind][r], src=r) for r in range(1, world)]
        else:
            works.append(dist.isend(band[kind], dst=0))
    if async_op:
        return _Works(works)
    for w in works:
        if w is not None and hasattr(w, "wait"):
            w.wait()
    return None

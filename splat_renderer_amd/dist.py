"""Multi-GPU frame: tile-row bands + ONE all-gather of projected splats.

The reference is single-device (no collective anywhere), so this module has no
reference counterpart; it composes the same C-ABI stages:

    rank r:  project splats [r*per, (r+1)*per)            splat_project_slice_compact
             all-gather the 16-byte exchange records       <- the frame's only exchange (RCCL/xGMI)
             my band's frame from the gathered records     splat_band_frame (record_format = COMPACT)

The exchange record is float4 {screen centre x, y, screen radius, depth}: the 32-byte ProjectedSplat is a
pure function of it and of the record's position in the gathered array (= global splat index), so half
the bytes cross xGMI and every rank rebuilds bounds bit-exactly where it needs them.  With the oriented-disc
footprint (HipStages(footprint="disc"): SequentialRenderer's splat) the record is 48 bytes — the 8-float disc
record, whose bounds are again a pure function of it, and the depth (splat_project_slice_disc).

Per-tile lists are the global stable order restricted to the tile, so the stitched image is
bit-identical to the single-GPU frame (tests/test_gpu_stages.py::test_band_rendering... on one GPU,
tests/test_dist_cpu.py for the sharding logic under gloo).

Process set-up: import torch BEFORE the first splat ctx is created (before libsplat_hip.so is loaded).  torch
bundles its own HIP runtime, and the one loaded first serves the whole process.

A second way to cut the same frame needs no exchange at all: every rank holds the splats anyway (the composite
gathers their colours), so it can project all of them itself and render its band (LocalBandRenderer:
splat_render_frame with tile_row0/1).  That costs each rank the full projection (75 us at 5M splats) instead of
1/world of it plus the all-gather; which is faster depends on the links, so bench.py times both and keeps the
faster.  The image is the same either way.

`stages` is the object that runs device work; the product always uses HipStages (below).  The
gloo CPU tests inject a checker-backed stand-in to exercise THIS file's slicing / gathering /
banding logic without a GPU — that stand-in lives under tests/, never here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import CompositeCfg, check

TILE = 16
REC_FLOATS = 4        # exchange record: centre x, centre y, screen radius, depth
DISC_REC_FLOATS = 12  # oriented-disc exchange record: disc record (8), depth, 3 x 0


def shard_size(n, world):
    return -(-n // world)


def slice_range(n, rank, world):
    """Contiguous index range projected by `rank`: global index = rank*per + local."""
    per = shard_size(n, world)
    first = min(rank * per, n)
    return first, min(per, n - first)


def band_rows(nty, rank, world):
    """Consecutive tile rows [r0, r1) composited by `rank`."""
    return nty * rank // world, nty * (rank + 1) // world


def balanced_rows(row_weights, world):
    """Band boundaries [r0, r1) per rank so that every band carries about the same weight (e.g. the
    tile-splat pairs of its rows) — the uniform-cube scene is denser in the centre rows.  Every rank
    computes this from the same all-reduced histogram, so all ranks agree; every band gets at least
    one row while rows remain.  Returns a list of (r0, r1)."""
    w = np.asarray(row_weights, dtype=np.float64)
    nty = w.shape[0]
    if world >= nty:
        return [(min(r, nty), min(r + 1, nty)) for r in range(world)]
    cum = np.concatenate([[0.0], np.cumsum(w + 1e-9)])  # strictly increasing
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        c = int(np.searchsorted(cum, total * r / world, side="left"))
        # nearest boundary to the ideal cut, keeping at least one row per band on both sides
        if c > 0 and abs(cum[c - 1] - total * r / world) < abs(cum[c] - total * r / world):
            c -= 1
        c = max(c, cuts[-1] + 1)
        c = min(c, nty - (world - r))
        cuts.append(c)
    cuts.append(nty)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def _project_slice(lib, ctx, disc, uniforms, props_ptr, normals_ptr, first, count, out_records):
    u = np.ascontiguousarray(uniforms, np.float32)
    up = u.ctypes.data_as(C.POINTER(C.c_float))
    if disc:  # the oriented disc lies in the tangent plane of the splat's normal
        if normals_ptr is None:
            raise ValueError("the oriented-disc projector needs the normals")
        check(lib.splat_project_slice_disc(ctx, up, props_ptr, 2, normals_ptr, 1, first, count, out_records.data_ptr()), ctx)
    else:
        check(lib.splat_project_slice_compact(ctx, up, props_ptr, 2, first, count, out_records.data_ptr()), ctx)


class HipStages:
    """Device work of one rank through the C ABI.  Tensors are torch CUDA tensors; the splat ctx
    is created on torch's current stream so that RCCL's stream dependencies order the exchange."""

    def __init__(self, torch, device_ordinal, n_total_padded, width, height, tile=TILE, mode=_lib.MODE_FRONT_TO_BACK,
                 early_out=True, footprint="isotropic"):
        self.torch = torch
        if footprint not in ("isotropic", "disc"):
            raise ValueError(f"footprint must be 'isotropic' or 'disc', not {footprint!r}")
        self.disc = footprint == "disc"
        self.rec_floats = DISC_REC_FLOATS if self.disc else REC_FLOATS
        self.lib = _lib.load()
        self.ordinal = device_ordinal
        p = C.c_void_p()
        stream = torch.cuda.current_stream().cuda_stream
        check(self.lib.splat_ctx_create_on_stream(device_ordinal, C.c_void_p(stream), C.byref(p)))
        self.ctx = p
        s = C.c_void_p()
        check(self.lib.splat_sort_create(self.ctx, n_total_padded, C.byref(s)), self.ctx)
        self.sorter = s
        b = C.c_void_p()
        check(self.lib.splat_bin_create(self.ctx, tile, C.byref(b)), self.ctx)
        self.binner = b
        self.width, self.height, self.tile = width, height, tile
        self.mode, self.early_out = mode, early_out
        self.pairs = 0
        self.overflows = 0
        self.misranked = 0  # frames whose lists failed the order check and were rendered again with ballots (SPLAT_ERR_RETRY)
        self.consumed = None  # optional torch int64[tiles, 2]: per tile, list entries {staged, consumed} by the composite
        self.lit = None       # optional torch float32[n,4]: lit colour plane (set_lit); band_frame then ignores props/normals
        self.pos_plane = None  # with it: the (pos, radius) plane, for local_frame
        self.local_projected = None  # local_frame: the projector's ProjectedSplat records

    def _again(self, rc):
        """Books a report about the previous sync-free frame: SPLAT_ERR_CAPACITY (pair limit) or SPLAT_ERR_RETRY (order check)."""
        if rc == _lib.ERR_RETRY:
            self.misranked += 1
        else:
            self.overflows += 1

    def rank_status(self):
        """Device.rankStatus() of this rank's context: {policy, atomicsOrdered, orderFaults}."""
        pol, ordered, faults = C.c_int(), C.c_int(), C.c_uint32()
        check(self.lib.splat_rank_status(self.ctx, C.byref(pol), C.byref(ordered), C.byref(faults)), self.ctx)
        return {"policy": ("checked", "atomic", "ballot")[pol.value], "atomicsOrdered": bool(ordered.value == 1),
                "orderFaults": int(faults.value)}

    def set_lit(self, props_ptr, normals_ptr, n):
        """Shade every splat once (kd from its normal) into a colour plane: the composite then gathers two
        lines per staged entry instead of three.  Call again when properties or normals change."""
        if self.lit is None or self.lit.shape[0] != n:
            self.lit = self.torch.empty((n, 4), dtype=self.torch.float32, device=f"cuda:{self.ordinal}")
        check(self.lib.splat_lit_colors(self.ctx, props_ptr + 16, 2, normals_ptr, 1, n, self.lit.data_ptr()), self.ctx)
        # the other plane of the native layout (what local_frame's projector reads next to the lit colours)
        if self.pos_plane is None or self.pos_plane.shape[0] != n:
            self.pos_plane = self.torch.empty((n, 4), dtype=self.torch.float32, device=f"cuda:{self.ordinal}")
        scratch = self.torch.empty((n, 4), dtype=self.torch.float32, device=f"cuda:{self.ordinal}")  # (the unlit colour plane)
        check(self.lib.splat_props_to_planes(self.ctx, props_ptr, n, self.pos_plane.data_ptr(), scratch.data_ptr()), self.ctx)
        check(self.lib.splat_sync(self.ctx), self.ctx)  # scratch is released on return

    def set_timing(self, enabled, stage_mask=0xFFFFFFFF):
        check(self.lib.splat_set_timing_stages(self.ctx, stage_mask), self.ctx)
        check(self.lib.splat_set_timing(self.ctx, int(enabled)), self.ctx)

    def stage_avg_ms(self, stage):
        cnt, tot = C.c_uint32(), C.c_double()
        check(self.lib.splat_stage_time_stats(self.ctx, stage, C.byref(cnt), C.byref(tot)), self.ctx)
        return tot.value / max(cnt.value, 1)

    def row_pairs(self):
        """Tile-splat pairs per tile row of the last band_frame (zeros outside the band)."""
        ntx, nty = -(-self.width // self.tile), -(-self.height // self.tile)
        counts = C.c_void_p()
        check(self.lib.splat_bin_counts(self.binner, C.byref(counts)), self.ctx)
        host = np.empty(ntx * nty, np.uint32)
        check(self.lib.splat_buf_download(self.ctx, host.ctypes.data, counts, host.nbytes), self.ctx)
        return host.reshape(nty, ntx).sum(axis=1).astype(np.int64)

    def new_records(self, count, fill_nan=False):
        t = self.torch.empty((count, self.rec_floats), dtype=self.torch.float32, device=f"cuda:{self.ordinal}")
        if fill_nan:
            t.fill_(float("nan"))
        return t

    def new_image(self):
        return self.torch.zeros((self.height, self.width, 4), dtype=self.torch.uint8, device=f"cuda:{self.ordinal}")

    def project_slice(self, uniforms, props_ptr, first, count, out_records, normals_ptr=None):
        _project_slice(self.lib, self.ctx, self.disc, uniforms, props_ptr, normals_ptr, first, count, out_records)

    def band_frame(self, records, n_records, props_ptr, normals_ptr, row0, row1, out_image, settle=False):
        """settle=False (frame loops): sync-free; a frame whose pairs outgrew 1.5x the previous frame's
        is only noticed at the next call (which then has room).  settle=True: wait for this frame's pair
        total and render it again if it overflowed — results are final on return."""
        prelit = self.lit is not None
        cfg = CompositeCfg(self.mode, int(self.early_out), self.tile, row0, row1,
                           _lib.RECORDS_DISC48 if self.disc else _lib.RECORDS_COMPACT, int(prelit),
                           _lib.FOOTPRINT_DISC if self.disc else _lib.FOOTPRINT_ISOTROPIC)
        if prelit:
            props_ptr, normals_ptr = self.lit.data_ptr(), None
        args = (self.ctx, self.sorter, self.binner, C.byref(cfg), props_ptr, normals_ptr, records.data_ptr(), n_records,
                self.width, self.height, out_image.data_ptr(), None,
                self.consumed.data_ptr() if self.consumed is not None else None)
        rc = self.lib.splat_band_frame(*args)
        if rc in _lib.RENDER_AGAIN:  # the PREVIOUS frame overflowed its sync-free limit (capacity was raised) or misranked: go again
            self._again(rc)
            rc = self.lib.splat_band_frame(*args)
        check(rc, self.ctx)
        if settle:
            t, k = C.c_uint64(), C.c_uint32()
            for _ in range(4):  # kept-count overflow, then pair overflow, at worst
                rc = self.lib.splat_band_settle(self.ctx, self.sorter, self.binner, C.byref(k), C.byref(t))
                if rc not in _lib.RENDER_AGAIN:
                    break
                self._again(rc)
                check(self.lib.splat_band_frame(*args), self.ctx)
            check(rc, self.ctx)
            self.pairs = int(t.value)

    def local_frame(self, uniforms, props_ptr, normals_ptr, n, row0, row1, out_image, settle=False):
        """Tile rows [row0, row1) of the frame from THIS rank's copy of the splats (splat_render_frame with a band): no
        exchange; every rank projects all n splats itself.  Same sync-free rules as band_frame."""
        prelit = self.lit is not None
        # (isotropic: the projector leaves lit composite records, one gathered line per staged entry)
        small = -(-self.width // self.tile) <= 256 and -(-self.height // self.tile) <= 256
        cfg = CompositeCfg(self.mode, int(self.early_out), self.tile, row0, row1,
                           _lib.RECORDS_LIT32 if (small and not self.disc) else _lib.RECORDS_PROJECTED, int(prelit),
                           _lib.FOOTPRINT_DISC if self.disc else _lib.FOOTPRINT_ISOTROPIC)
        if self.local_projected is None or self.local_projected.shape[0] < n:
            self.local_projected = self.torch.empty((max(n, 1), 8), dtype=self.torch.float32, device=f"cuda:{self.ordinal}")
        u = np.ascontiguousarray(uniforms, np.float32)
        head = (self.ctx, self.sorter, self.binner, C.byref(cfg), u.ctypes.data_as(C.POINTER(C.c_float)))
        tail = (normals_ptr if (self.disc or not prelit) else None, n, self.width, self.height, self.local_projected.data_ptr(),
                out_image.data_ptr(), None)
        if prelit:
            fn, args = self.lib.splat_render_frame_planes, head + (self.pos_plane.data_ptr(), self.lit.data_ptr()) + tail
        else:
            fn, args = self.lib.splat_render_frame, head + (props_ptr,) + tail
        rc = fn(*args)
        if rc in _lib.RENDER_AGAIN:  # the PREVIOUS frame overflowed its sync-free limit (capacity was raised) or misranked: go again
            self._again(rc)
            rc = fn(*args)
        check(rc, self.ctx)
        if settle:
            t = C.c_uint64()
            rc = self.lib.splat_bin_total(self.binner, C.byref(t))
            if rc in _lib.RENDER_AGAIN:
                self._again(rc)
                check(fn(*args), self.ctx)
                rc = self.lib.splat_bin_total(self.binner, C.byref(t))
            check(rc, self.ctx)
            self.pairs = int(t.value)

    def timing_consumed(self):
        """List entries (staged, consumed) by the composites of local_frame since timing was switched on."""
        s, c = C.c_uint64(), C.c_uint64()
        check(self.lib.splat_timing_consumed(self.ctx, C.byref(s), C.byref(c)), self.ctx)
        return int(s.value), int(c.value)

    @property
    def kept(self):
        """Splats the last band_frame kept (synchronises)."""
        k = C.c_uint32()
        check(self.lib.splat_band_kept(self.ctx, self.sorter, C.byref(k)), self.ctx)
        return k.value

    def destroy(self):
        self.lib.splat_bin_destroy(self.binner)
        self.lib.splat_sort_destroy(self.sorter)
        self.lib.splat_ctx_destroy(self.ctx)


class AbiAllGather:
    """The frame's exchange through the C ABI's own RCCL communicator (splat_comm_init / splat_allgather_records) — the
    path a host without torch.distributed uses (napi/index.js: Comm) — as an `all_gather(out, shard)` callable for
    BandRenderer / FramePipeline.  The unique id travels from rank 0 by `broadcast_bytes(numpy uint8[128]) -> same
    array on every rank` (torch.distributed.broadcast in bench.py; any channel will do).  The collective is enqueued
    on the splat ctx bound to the torch stream that is current at the call (register every ctx that exchanges)."""

    def __init__(self, torch, stages, rank, world, broadcast_bytes):
        self.torch, self.lib, self.rank, self.world = torch, stages.lib, rank, world
        ident = np.zeros(_lib.COMM_ID_BYTES, np.uint8)
        if rank == 0:
            check(self.lib.splat_comm_unique_id(ident.ctypes.data))
        ident = np.ascontiguousarray(broadcast_bytes(ident), np.uint8)
        c = C.c_void_p()
        check(self.lib.splat_comm_init(stages.ctx, rank, world, ident.ctypes.data, C.byref(c)), stages.ctx)
        self.comm = c
        self.ctx_of_stream = {}
        self.register(torch.cuda.current_stream(), stages.ctx)

    def register(self, stream, ctx):
        self.ctx_of_stream[int(stream.cuda_stream)] = ctx

    def __call__(self, out, shard):
        ctx = self.ctx_of_stream[int(self.torch.cuda.current_stream().cuda_stream)]
        check(self.lib.splat_allgather_records(ctx, self.comm, shard.data_ptr(), out.data_ptr(), shard.numel() * shard.element_size()), ctx)

    def rccl_view(self):
        """(ranks, rank) as the communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
        n, k = C.c_int(), C.c_int()
        check(self.lib.splat_comm_count(self.comm, C.byref(n), C.byref(k)))
        return int(n.value), int(k.value)

    def destroy(self):
        if self.comm:
            self.lib.splat_comm_destroy(self.comm)
            self.comm = None


class BandRenderer:
    """One rank of a multi-GPU frame.  `all_gather(out, shard)` is AbiAllGather (the C ABI's RCCL communicator) or
    torch.distributed.all_gather_into_tensor (RCCL on GPUs; gloo in the CPU tests)."""

    def __init__(self, stages, n, width, height, rank, world, all_gather, tile=TILE, gathered=None, always_gather=False):
        # gathered: a records tensor to gather into instead of one of this renderer's own (virtual ranks on one device share it)
        # always_gather: issue the collective with one rank too (bench.py --band-path: the N > 1 path rehearsed on one GPU)
        self.stages, self.n, self.rank, self.world = stages, n, rank, world
        self.exchanging = world > 1 or always_gather
        self.width, self.height, self.tile = width, height, tile
        self.per = shard_size(n, world)
        self.first, self.count = slice_range(n, rank, world)
        self.nty = -(-height // tile)
        self.row0, self.row1 = band_rows(self.nty, rank, world)
        self.all_gather = all_gather
        # shard padding (indices >= n) is all-NaN once: NaN bins nowhere and never changes
        self.shard = stages.new_records(self.per, fill_nan=True)
        self.gathered = gathered if gathered is not None else stages.new_records(self.per * world) if self.exchanging else self.shard
        self.image = stages.new_image()

    def render(self, uniforms, props_ptr, normals_ptr, settle=False):
        st = self.stages
        st.project_slice(uniforms, props_ptr, self.first, self.count, self.shard, normals_ptr)
        if self.exchanging:
            self.all_gather(self.gathered, self.shard)
        if settle:
            st.band_frame(self.gathered, self.per * self.world, props_ptr, normals_ptr, self.row0, self.row1, self.image, True)
        else:
            st.band_frame(self.gathered, self.per * self.world, props_ptr, normals_ptr, self.row0, self.row1, self.image)
        return self.image

    def verify_exchange(self, uniforms, props_ptr, normals_ptr=None, include_self=False):
        """Self-check of the frame's exchange (every rank holds all splats, so it can recompute any shard): projects each
        OTHER rank's slice locally and compares it, bit for bit, with the block that rank contributed to the last
        all-gather (render() with the same uniforms first).  Returns the number of other ranks whose block matched.
        include_self: this rank's own block too (a one-rank rehearsal has no other)."""
        st, torch = self.stages, self.stages.torch
        tmp = st.new_records(self.per, fill_nan=True)
        rec = self.gathered.reshape(self.world, self.per, -1)
        ok = 0
        for r in range(self.world):
            if r == self.rank and not include_self:
                continue
            first, count = slice_range(self.n, r, self.world)
            st.project_slice(uniforms, props_ptr, first, count, tmp, normals_ptr)
            if hasattr(torch, "cuda") and tmp.is_cuda:
                torch.cuda.synchronize()
            ok += int(torch.equal(tmp[:count].view(torch.int32), rec[r, :count].view(torch.int32)))
        return ok

    def rebalance(self, all_reduce_sum):
        """Re-cut the bands from the pairs-per-row histogram of the frame just rendered.
        all_reduce_sum(tensor) sums an int64 tensor over ranks in place (torch.distributed.all_reduce)."""
        rows = self.stages.row_pairs()
        t = self.stages.torch.from_numpy(rows)
        if self.world > 1:
            t = t.to(self.gathered.device)
            all_reduce_sum(t)
            rows = t.cpu().numpy()
        self.row0, self.row1 = balanced_rows(rows, self.world)[self.rank]
        return rows

    def pixel_rows(self):
        return self.row0 * self.tile, min(self.row1 * self.tile, self.height)


class LocalBandRenderer:
    """One rank of a multi-GPU frame with no exchange: the rank projects all n splats from its own copy and renders
    its band of tile rows.  Same surface as BandRenderer where bench.py and the tests use it."""

    def __init__(self, stages, n, width, height, rank, world, tile=TILE):
        self.stages, self.n, self.rank, self.world = stages, n, rank, world
        self.width, self.height, self.tile = width, height, tile
        self.nty = -(-height // tile)
        self.row0, self.row1 = band_rows(self.nty, rank, world)
        self.image = stages.new_image()

    def render(self, uniforms, props_ptr, normals_ptr, settle=False):
        self.stages.local_frame(uniforms, props_ptr, normals_ptr, self.n, self.row0, self.row1, self.image, settle)
        return self.image

    def pixel_rows(self):
        return self.row0 * self.tile, min(self.row1 * self.tile, self.height)


class ProjectStage:
    """The projector alone, on its own splat ctx bound to a second stream: the exchange side of a
    pipelined multi-GPU frame loop (FramePipeline)."""

    def __init__(self, torch, device_ordinal, stream, disc=False):
        self.lib = _lib.load()
        self.disc = disc
        p = C.c_void_p()
        check(self.lib.splat_ctx_create_on_stream(device_ordinal, C.c_void_p(stream.cuda_stream), C.byref(p)))
        self.ctx = p

    def project_slice(self, uniforms, props_ptr, first, count, out_records, normals_ptr=None):
        _project_slice(self.lib, self.ctx, self.disc, uniforms, props_ptr, normals_ptr, first, count, out_records)

    def destroy(self):
        self.lib.splat_ctx_destroy(self.ctx)


class FramePipeline:
    """Two frames in flight on one rank: while the band of frame k is binned, sorted and composited on the
    main stream, frame k+1's slice is projected and all-gathered on a second stream (RCCL overlapped with
    compute).  Frames are unchanged; throughput becomes 1 / max(exchange, band work) instead of
    1 / (exchange + band work).  Two (shard, gathered) buffer pairs alternate; two events per pair order
    the streams: `ready` (exchange done -> the band frame may read) and `free` (band frame done -> the next
    exchange may overwrite)."""

    def __init__(self, torch, br, device_ordinal):
        self.torch, self.br = torch, br
        self.main = torch.cuda.current_stream()
        self.comm = torch.cuda.Stream()
        self.proj = ProjectStage(torch, device_ordinal, self.comm, getattr(br.stages, "disc", False))
        if hasattr(br.all_gather, "register"):  # AbiAllGather: the exchange runs on the second stream's ctx
            br.all_gather.register(self.comm, self.proj.ctx)
        st = br.stages
        self.shards = [br.shard, st.new_records(br.per, fill_nan=True)]
        self.gathered = [br.gathered, st.new_records(br.per * br.world) if br.exchanging else self.shards[1]]
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.free = [torch.cuda.Event(), torch.cuda.Event()]
        self.used = [False, False]

    def exchange(self, k, uniforms, props_ptr, normals_ptr=None):
        """Project and all-gather frame k's records (asynchronous, on the second stream)."""
        s, br = k & 1, self.br
        with self.torch.cuda.stream(self.comm):
            if self.used[s]:
                self.comm.wait_event(self.free[s])  # the band frame that read this pair has finished
            self.proj.project_slice(uniforms, props_ptr, br.first, br.count, self.shards[s], normals_ptr)
            if br.exchanging:
                br.all_gather(self.gathered[s], self.shards[s])
            self.ready[s].record(self.comm)

    def band(self, k, props_ptr, normals_ptr, settle=False):
        """Frame k's band from its gathered records (main stream)."""
        s, br = k & 1, self.br
        self.main.wait_event(self.ready[s])
        br.stages.band_frame(self.gathered[s], br.per * br.world, props_ptr, normals_ptr, br.row0, br.row1, br.image, settle)
        self.free[s].record(self.main)
        self.used[s] = True
        return br.image

    def run(self, frames, uniforms_of, props_ptr, normals_ptr):
        """frames frames; uniforms_of(k) gives frame k's uniform block."""
        self.exchange(0, uniforms_of(0), props_ptr, normals_ptr)
        for k in range(frames):
            if k + 1 < frames:
                self.exchange(k + 1, uniforms_of(k + 1), props_ptr, normals_ptr)
            self.band(k, props_ptr, normals_ptr)

    def destroy(self):
        self.torch.cuda.synchronize()
        self.proj.destroy()

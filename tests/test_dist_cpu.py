"""Multi-GPU path on CPU: world_size-2 (and 3) gloo runs of splat_renderer_amd.dist.BandRenderer
with a checker-backed stand-in for the device stages (the stand-in is test infrastructure; the
product always uses dist.HipStages).  What is tested is dist.py's own logic — slice ranges, NaN
shard padding, the one all-gather, band rows, stitching — against the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import oracle as O  # noqa: E402
from splat_renderer_amd import dist  # noqa: E402
from tests.helpers import make_case, oracle_pipeline  # noqa: E402


class OracleStages:
    """Same surface as dist.HipStages, computed by the oracle on CPU tensors."""

    def __init__(self, width, height, tile=16, disc=False):
        self.width, self.height, self.tile = width, height, tile
        self.kept = 0
        self.torch = torch
        self.disc = disc  # the oriented-disc footprint: 48-byte exchange records {disc record, depth, 0, 0, 0}

    def new_records(self, count, fill_nan=False):
        t = torch.zeros((count, 12 if self.disc else 4), dtype=torch.float32)  # the 16-byte (48-byte) exchange records
        if fill_nan:
            t.fill_(float("nan"))
        return t

    def new_image(self):
        return torch.zeros((self.height, self.width, 4), dtype=torch.uint8)

    def project_slice(self, uniforms, props, first, count, out_records, normals=None):
        if self.disc:
            proj, discs = O.project_disc(uniforms, props[first:first + count], normals[first:first + count])
            out_records[:count, :8] = torch.from_numpy(discs)
            out_records[:count, 8] = torch.from_numpy(proj[:, 4].copy())
            out_records[:count, 9:] = 0
            return
        out_records[:count] = torch.from_numpy(O.project_compact(uniforms, props[first:first + count]))

    def local_frame(self, uniforms, props, normals, n, row0, row1, out_image, settle=False):
        """dist.LocalBandRenderer's stage: the band from this rank's own copy of all n splats (no exchange)."""
        rec = self.new_records(n)
        self.project_slice(uniforms, props, 0, n, rec, normals)
        self.band_frame(rec, n, props, normals, row0, row1, out_image)

    def _records_of(self, records):
        """ProjectedSplat records (originalIndex = position = global index) rebuilt from the exchange records."""
        if not self.disc:
            return O.expand_compact(records.numpy())
        r48 = records.numpy()
        rec = np.zeros((r48.shape[0], 8), np.float32)
        for i in range(r48.shape[0]):
            ok, b = O.disc_bounds(r48[i, :8])  # NaN padding -> not ok -> zeros -> bins nowhere
            rec[i, :4] = b
        rec[:, 4] = r48[:, 8]
        return rec

    def band_frame(self, records, n_records, props, normals, row0, row1, out_image, settle=False):
        rec = self._records_of(records)
        ntx, nty = -(-self.width // self.tile), -(-self.height // self.tile)
        # splat_band_keys: keep splats whose clamped tile rows meet [row0,row1), ascending index
        keep = []
        for i in range(n_records):
            b = rec[i]
            if np.isnan(b[:4]).any():
                continue
            mny, mxy = max(b[1], 0.0), min(b[3], float(self.height))
            mnx, mxx = max(b[0], 0.0), min(b[2], float(self.width))
            if mnx >= mxx or mny >= mxy:
                continue
            ty0, ty1 = int(mny // self.tile), min(int(mxy // self.tile), nty - 1)
            if max(ty0, row0) <= min(ty1, row1 - 1):
                keep.append(i)
        keep = np.array(keep, np.uint32)
        self.kept = keep.shape[0]
        keys, _ = O.extract_keys(rec[keep] if keep.size else np.zeros((0, 8), np.float32))
        order = keep[np.argsort(keys, kind="stable")] if keep.size else keep
        counts, offsets, idx = O.bin_sorted(rec, order, self.width, self.height, self.tile)
        # restrict lists to the band's rows (splat_bin_run's tile_row0/1)
        c2 = counts.reshape(nty, ntx).copy()
        c2[:row0] = 0
        c2[row1:] = 0
        lists = [idx[offsets[t]:offsets[t] + counts[t]] if c2.reshape(-1)[t] else idx[:0] for t in range(ntx * nty)]
        counts = c2.reshape(-1).astype(np.uint32)
        offsets, _ = O.scan_exclusive(counts)
        idx = np.concatenate(lists) if lists else idx[:0]
        r0, r1 = row0 * self.tile, min(row1 * self.tile, self.height)
        if self.disc:
            discs = np.ascontiguousarray(records.numpy()[:, :8])
            _, img8, _, _ = O.composite_disc(True, props[:, 4:], normals, discs, idx, counts, offsets, self.width, self.height,
                                             self.tile, rows=(r0, r1))
        else:
            _, img8, _ = O.composite(O.MODE_FRONT_TO_BACK, True, props[:, 4:], normals, rec[:, :8], idx, counts, offsets,
                                     self.width, self.height, self.tile, rows=(r0, r1))
        out_image[r0:r1] = torch.from_numpy(img8[r0:r1])


def _worker(rank, world, port, n, w, h, out_dir, disc=False, local=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    props, normals, u = make_case(n, w, h, 31, 1.5)
    st = OracleStages(w, h, disc=disc)
    br = (dist.LocalBandRenderer(st, n, w, h, rank, world) if local else
          dist.BandRenderer(st, n, w, h, rank, world, td.all_gather_into_tensor))
    img = br.render(u, props, normals)
    if not local:  # bench.py's self-check of the exchange: every other rank's gathered shard equals this rank's projection of it
        assert br.verify_exchange(u, props, normals) == world - 1
    r0, r1 = br.pixel_rows()
    np.save(os.path.join(out_dir, f"band{rank}.npy"), img.numpy()[r0:r1])
    np.save(os.path.join(out_dir, f"rows{rank}.npy"), np.array([r0, r1, st.kept]))
    td.barrier()
    td.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_band_renderer_gloo_matches_single_process(world, tmp_path):
    n, w, h = 1500, 160, 112  # 7 tile rows: uneven bands; n not divisible by 2 or 3 -> NaN shard padding
    n += 1
    props, normals, u = make_case(n, w, h, 31, 1.5)
    a = oracle_pipeline(props, normals, u, w, h)
    _, want8, _ = O.composite(O.MODE_FRONT_TO_BACK, True, props[:, 4:], normals, a["proj"], a["indices"], a["counts"],
                              a["offsets"], w, h)
    mp.spawn(_worker, args=(world, _free_port(), n, w, h, str(tmp_path)), nprocs=world, join=True)
    got = np.zeros_like(want8)
    covered = 0
    kept_total = 0
    for r in range(world):
        r0, r1, kept = np.load(tmp_path / f"rows{r}.npy")
        got[r0:r1] = np.load(tmp_path / f"band{r}.npy")
        covered += r1 - r0
        kept_total += kept
    assert covered == h
    assert np.array_equal(got, want8)  # bit-identical to the single-process frame
    assert kept_total >= (a["counts"].reshape(7, 10).sum(axis=1) > 0).sum()  # every non-empty row has an owner


def test_band_renderer_gloo_oriented_disc_footprint(tmp_path):
    """The same sharding with the oriented-disc footprint's 48-byte exchange records (world 2)."""
    world, n, w, h = 2, 1201, 160, 112
    props, normals, u = make_case(n, w, h, 31, 1.5)
    proj, discs = O.project_disc(u, props, normals)
    keys, pay = O.extract_keys(proj)
    _, order = O.sort_pairs(keys, pay)
    counts, offsets, idx = O.bin_sorted(proj, order, w, h)
    _, want8, _, _ = O.composite_disc(True, props[:, 4:], normals, discs, idx, counts, offsets, w, h)
    mp.spawn(_worker, args=(world, _free_port(), n, w, h, str(tmp_path), True), nprocs=world, join=True)
    got = np.zeros_like(want8)
    for r in range(world):
        r0, r1, _ = np.load(tmp_path / f"rows{r}.npy")
        got[r0:r1] = np.load(tmp_path / f"band{r}.npy")
    assert np.array_equal(got, want8)


def test_local_band_renderer_gloo_no_exchange(tmp_path):
    """dist.LocalBandRenderer under gloo world 2: every rank renders its band from its own copy; no collective."""
    world, n, w, h = 2, 1201, 160, 112
    props, normals, u = make_case(n, w, h, 31, 1.5)
    a = oracle_pipeline(props, normals, u, w, h)
    _, want8, _ = O.composite(O.MODE_FRONT_TO_BACK, True, props[:, 4:], normals, a["proj"], a["indices"], a["counts"],
                              a["offsets"], w, h)
    mp.spawn(_worker, args=(world, _free_port(), n, w, h, str(tmp_path), False, True), nprocs=world, join=True)
    got = np.zeros_like(want8)
    for r in range(world):
        r0, r1, _ = np.load(tmp_path / f"rows{r}.npy")
        got[r0:r1] = np.load(tmp_path / f"band{r}.npy")
    assert np.array_equal(got, want8)


def test_slice_and_band_partition_properties():
    for n in (0, 1, 7, 1000, 5_000_000):
        for world in (1, 2, 3, 8):
            per = dist.shard_size(n, world) if n else 0
            cover = []
            for r in range(world):
                first, count = dist.slice_range(n, r, world)
                assert 0 <= count <= per and first + count <= n
                cover.append((first, count))
            assert sum(c for _, c in cover) == n
            assert all(cover[i][0] + cover[i][1] == cover[i + 1][0] or cover[i + 1][1] == 0 for i in range(world - 1))
    for nty in (1, 7, 68, 135):
        for world in (1, 2, 4, 8):
            rows = [dist.band_rows(nty, r, world) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == nty
            assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))


def test_balanced_rows():
    w = np.array([1, 1, 10, 30, 30, 10, 1, 1], np.int64)
    for world in (1, 2, 3, 4, 8, 12):
        bands = dist.balanced_rows(w, world)
        assert len(bands) == world and bands[0][0] == 0 and bands[-1][1] == 8 or world > 8
        flat = [r for a, b in bands for r in range(a, b)]
        assert flat == sorted(set(flat)) and set(flat) == set(range(8))  # a partition of the rows
        if world <= 8:
            assert all(b > a for a, b in bands)  # nobody idles while rows remain
    two = dist.balanced_rows(w, 2)
    assert two == [(0, 4), (4, 8)]
    sums = [w[a:b].sum() for a, b in dist.balanced_rows(w, 4)]
    assert max(sums) <= 42  # equal rows would give 2/40/40/2; the balanced cut does much better
    assert dist.balanced_rows(np.zeros(5), 2) in ([(0, 2), (2, 5)], [(0, 3), (3, 5)])

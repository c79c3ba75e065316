#!/usr/bin/env python3
"""Generates tests/golden/ref_binsorted_*.npz and ref_scan.npz — OUTPUTS OF THE REFERENCE'S OWN CODE.

Almost all of the reference's hot path needs a WebGPU device, but two pieces are plain
CPU loops: the count / exclusive-scan / fill loops of `TileBinner.binSorted` (the definition of the
tile lists, parity contract 2) and the loop of `PrefixSumScanner.scanCPU`.  This script, run in the
build container (the reference is not present anywhere else), reads those statements from
/root/reference as text, strips TypeScript-only syntax IN MEMORY (there is none in these ranges beyond
non-null assertions, but the stripper is applied anyway), and executes them under the container's
Node 12 — the program goes to `node -` on stdin, nothing of it is written to disk — on the inputs of
the committed fixtures plus an edge-case set (off-screen, straddling, NaN, padding indices).  Only the
resulting arrays are stored.  tests/ then hold BOTH the oracle and the HIP path to these arrays, so the
integer half of the parity contract (tile counts, offsets, lists; the scan) is pinned to an execution
of the reference itself.  Still unpinned: every float stage (projector, composite) and gl-matrix.

Run from the repo root:  python tests/golden/make_ref_fixtures.py
"""
import json
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
TILE = 16


def strip_types(js):
    """Erasable TypeScript syntax -> JavaScript, for statement bodies (no declarations with generics in here)."""
    js = re.sub(r"(\b(?:let|const|var)\s+\w+)\s*:\s*[\w<>\[\]| ]+(?=\s*=)", r"\1", js)  # let x: T = ...
    js = re.sub(r"(\w|\))!(?=[.\[,;)\s])", r"\1", js)                                   # non-null assertion x!
    js = re.sub(r"\s+as\s+\w+(\[\])?", "", js)                                          # casts
    return js


def between(text, first, last, include_last=False):
    a = text.index(first)
    b = text.index(last, a)
    return text[a:b + (len(last) if include_last else 0)]


def reference_statements():
    tb = open(os.path.join(REF, "TileBinner.ts")).read()
    body = between(tb, "async binSorted(", "async bin(")
    # src/TileBinner.ts:426-495: from the count pass to the end of the fill pass
    loops = between(body, "this.tileCounts = new Uint32Array(this.numTiles);", "readbackBuffer.unmap();")
    ps = open(os.path.join(REF, "PrefixSumScanner.ts")).read()
    scan_body = between(ps, "private async scanCPU(", "cleanupTempBuffers(): void")
    # src/PrefixSumScanner.ts:150-155
    scan = between(scan_body, "const outputData = new Uint32Array(numElements);", "readbackBuffer.unmap();")
    return strip_types(loops), strip_types(scan)


PROGRAM = """
const input = JSON.parse(require('fs').readFileSync(0, 'utf8'));
const GPUBufferUsage = { STORAGE: 0, COPY_DST: 0, MAP_READ: 0 };
function binSorted(sortedIndices, projectedData, screenWidth, screenHeight) {
  %(LOOPS)s
  return { counts: Array.from(this.tileCounts), offsets: Array.from(this.tileOffsets), indices: Array.from(indices),
           total: this.totalSplatCount };
}
function scanCPU(inputData, numElements) {
  %(SCAN)s
  return Array.from(outputData);
}
const out = {};
for (const name of Object.keys(input.bins)) {
  const c = input.bins[name];
  const ntx = Math.ceil(c.width / c.tile), nty = Math.ceil(c.height / c.tile);   // TileBinner.ts ensureBuffers
  const self = { numTiles: ntx * nty, numTilesX: ntx, numTilesY: nty, tileSize: c.tile, splatIndicesBuffer: null,
                 device: { createBuffer: () => ({ destroy() {} }) } };
  const projected = new Float32Array(new Uint32Array(c.projected_bits).buffer);
  out[name] = binSorted.call(self, new Uint32Array(c.sorted), projected, c.width, c.height);
}
out.scans = input.scans.map(a => scanCPU(new Uint32Array(a), a.length));
process.stdout.write(JSON.stringify(out));
"""


def edge_case():
    """Bounds chosen by hand: off-screen on every side, straddling corners, exactly on tile edges, degenerate,
    NaN in every position, and padding indices (0xFFFFFFFF) in the sorted order."""
    w, h = 70, 52  # ragged: 5 x 4 tiles
    nan = np.nan
    b = [[-50, 10, -20, 30], [10, 70, 30, 90], [-5, -5, 5, 5], [60, 40, 100, 100], [16, 16, 32, 32], [nan, 0, 10, 10],
         [0, nan, 10, 10], [0, 0, nan, 10], [0, 0, 10, nan], [20, 20, 20, 40], [20, 20, 40, 20], [30, 30, 10, 10],
         [0, 0, 70, 52], [69.5, 51.5, 69.9, 51.9], [-1e9, -1e9, 1e9, 1e9], [70, 0, 80, 10], [0, 52, 10, 60],
         [15.999, 15.999, 16.0, 16.0], [31.5, 0.25, 48.0, 15.75], [np.inf, 0, np.inf, 10], [-np.inf, 0, np.inf, 10]]
    rec = np.zeros((len(b), 8), np.float32)
    rec[:, :4] = np.array(b, np.float32)
    order = np.array([3, 0xFFFFFFFF, 4, 2, 1, 0, 5, 0xFFFFFFFF, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6], np.uint32)
    return rec, order, w, h


def main():
    loops, scan = reference_statements()
    program = PROGRAM % {"LOOPS": loops, "SCAN": scan}
    bins, inputs = {}, {}
    for name in ("tiny7", "small300", "ragged1000"):
        g = np.load(os.path.join(HERE, name + ".npz"))
        n, w, h, _ = (int(x) for x in g["dims"])
        inputs[name] = (g["projected"], g["order"][:n].copy(), w, h)
    rec, order, w, h = edge_case()
    inputs["edges"] = (rec, order, w, h)
    for name, (proj, order, w, h) in inputs.items():
        bins[name] = {"width": w, "height": h, "tile": TILE, "sorted": [int(x) for x in order],
                      "projected_bits": [int(x) for x in np.ascontiguousarray(proj, np.float32).view(np.uint32).reshape(-1)]}
    rng = np.random.default_rng(11)
    scans = [[1, 2, 3, 4, 5], [0] * 9, [7], [int(x) for x in rng.integers(0, 5000, 8160)], [int(x) for x in rng.integers(0, 3, 1024)]]
    # the program is handed to node as an argument, the data on stdin: nothing of either is written to disk
    r = subprocess.run(["node", "-e", program], input=json.dumps({"bins": bins, "scans": scans}), capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit("node failed:\n" + r.stderr)
    out = json.loads(r.stdout)
    for name, (proj, order, w, h) in inputs.items():
        o = out[name]
        assert o["total"] == len(o["indices"])
        np.savez_compressed(os.path.join(HERE, f"ref_binsorted_{name}.npz"), projected=np.ascontiguousarray(proj, np.float32),
                            sorted=order, dims=np.array([w, h, TILE], np.int64), counts=np.array(o["counts"], np.uint32),
                            offsets=np.array(o["offsets"], np.uint32), indices=np.array(o["indices"], np.uint32))
        print(f"ref_binsorted_{name}.npz: {len(o['counts'])} tiles, {o['total']} pairs")
    np.savez_compressed(os.path.join(HERE, "ref_scan.npz"), **{f"in{i}": np.array(a, np.uint32) for i, a in enumerate(scans)},
                        **{f"out{i}": np.array(a, np.uint32) for i, a in enumerate(out["scans"])})
    print(f"ref_scan.npz: {len(scans)} cases; node {subprocess.run(['node', '--version'], capture_output=True, text=True).stdout.strip()}")


if __name__ == "__main__":
    main()

"""Pure-Python model of the merge-path SpMV kernels in pygrank_amd/csrc/pgh_spmv.hip / pgh_graph.hip.

Transcribes the device logic step by step (tile table, per-thread merge-path search, serial walk, shuffle
segmented scan, cross-wavefront hand-off, tile carries, chain_first, fix-up) with configurable workgroup /
wavefront sizes so that tiny inputs exercise every seam.  Used by tests/test_merge_path_model.py to validate
the algorithm on the CPU before it is trusted on the GPU.
"""
import numpy as np

SENTINEL = 0x7fffffff


def tile_coords(rowptr, n, nnz, items):
    total = n + nnz
    num_tiles = (total + items - 1) // items
    coord = []
    for t in range(num_tiles + 1):
        d = min(t * items, total)
        lo, hi = max(d - nnz, 0), min(d, n)
        while lo < hi:
            mid = (lo + hi) >> 1
            if rowptr[mid + 1] <= d - mid - 1:
                lo = mid + 1
            else:
                hi = mid
        coord.append((lo, d - lo))
    return num_tiles, coord


def chain_first(rowptr, coord, n, num_tiles):
    out = []
    for t in range(num_tiles):
        row0, z0 = coord[t]
        first = -1
        if row0 < n and rowptr[row0] < z0 and coord[t + 1][0] > row0:
            s = t - 1
            while s > 0 and coord[s][0] == row0:
                s -= 1
            first = s
        out.append(first)
    return out


def spmv_model(rowptr, col, val, x, ipt=3, wg=16, wave=4):
    """Returns y = A x for CSR (rowptr, col, val) following the kernel's exact control flow."""
    n, nnz = len(rowptr) - 1, len(col)
    items = wg * ipt
    num_tiles, coord = tile_coords(rowptr, n, nnz, items)
    cfirst = chain_first(rowptr, coord, n, num_tiles)
    tail_carry = np.zeros(max(num_tiles, 1))
    head_partial = np.zeros(max(num_tiles, 1))
    y = np.full(n, np.nan)
    nwaves = wg // wave
    for tile in range(num_tiles):
        row0, z0 = coord[tile]
        row1, z1 = coord[tile + 1]
        tile_rows, tile_nnz = row1 - row0, z1 - z0
        tile_items = tile_rows + tile_nnz
        assert tile_items <= items
        s_rend = [(rowptr[row0 + r + 1] - z0) if row0 + r < n else SENTINEL for r in range(tile_rows + 1)]
        s_prod = [val[z0 + k] * x[col[z0 + k]] for k in range(tile_nnz)]
        s_rsum = [None] * max(tile_rows, 1)
        keys, vals, first_emit, first_val = [0] * wg, [0.0] * wg, [-1] * wg, [0.0] * wg
        for tid in range(wg):
            d0 = min(tid * ipt, tile_items)
            d1 = min(d0 + ipt, tile_items)
            lo, hi = max(d0 - tile_nnz, 0), min(d0, tile_rows)
            while lo < hi:
                mid = (lo + hi) >> 1
                if s_rend[mid] <= d0 - mid - 1:
                    lo = mid + 1
                else:
                    hi = mid
            i, j = lo, d0 - lo
            acc = 0.0
            rend = s_rend[i]
            for k in range(ipt):
                if d0 + k < d1:
                    if j < rend:
                        acc += s_prod[j]
                        j += 1
                    else:
                        if first_emit[tid] < 0:
                            first_emit[tid], first_val[tid] = i, acc
                        else:
                            s_rsum[i] = acc
                        acc = 0.0
                        i += 1
                        rend = s_rend[i]
            keys[tid], vals[tid] = i, acc
        # wave-level segmented inclusive scan (Hillis-Steele with key equality)
        for w in range(nwaves):
            base = w * wave
            off = 1
            while off < wave:
                nk = list(keys[base:base + wave])
                nv = list(vals[base:base + wave])
                for lane in range(wave):
                    if lane >= off and keys[base + lane - off] == keys[base + lane]:
                        nv[lane] = vals[base + lane] + vals[base + lane - off]
                vals[base:base + wave] = nv
                off <<= 1
        wkey = [keys[w * wave + wave - 1] for w in range(nwaves)]
        wval = [vals[w * wave + wave - 1] for w in range(nwaves)]
        pks, pvs = [], []
        for w in range(nwaves):
            pk, pv = -1, 0.0
            for w2 in range(w):
                if wkey[w2] == pk:
                    pv += wval[w2]
                else:
                    pk, pv = wkey[w2], wval[w2]
            pks.append(pk)
            pvs.append(pv)
        incl = list(vals)
        for tid in range(wg):
            w = tid // wave
            if pks[w] == keys[tid]:
                incl[tid] = vals[tid] + pvs[w]
        for tid in range(wg):
            w, lane = tid // wave, tid % wave
            if lane == 0:
                ek, ev = pks[w], pvs[w]
            else:
                ek, ev = keys[tid - 1], incl[tid - 1]
            if first_emit[tid] >= 0:
                total = first_val[tid] + (ev if (tid > 0 and ek == first_emit[tid]) else 0.0)
                s_rsum[first_emit[tid]] = total
                if first_emit[tid] == 0:
                    head_partial[tile] = total
        tail_carry[tile] = incl[wg - 1]
        head_spans = row0 < n and rowptr[row0] < z0
        for r in range(tile_rows):
            if r == 0 and head_spans:
                continue
            assert s_rsum[r] is not None, (tile, r)
            assert np.isnan(y[row0 + r])
            y[row0 + r] = s_rsum[r]
    # fix-up
    for t in range(num_tiles):
        first = cfirst[t]
        if first < 0:
            continue
        total = sum(tail_carry[s] for s in range(first, t)) + head_partial[t]
        row = coord[t][0]
        assert np.isnan(y[row]), ("row written twice", row)
        y[row] = total
    assert not np.isnan(y).any(), "some row was never written"
    return y

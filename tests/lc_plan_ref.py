"""numpy restatement of the LaneConv work-item plan (lgcn_lc_plan_build, csrc/lgcn_laneconv.hip: k_lc_plan).

Test infrastructure: the device plan is an integer structure and must match this one bit for bit.  It also gives
the CPU suite something to check without a GPU: that a plan covers every edge of the CSR exactly once."""
import numpy as np

LC_UNITS = 15
HDR = 8


def csr_row(rowptr, n_rel, n, r):
    k = ((n >> 4) * n_rel + r) * 16 + (n & 15)
    return int(rowptr[k]), int(rowptr[k + 1])


def lc_plan_ref(rowptr, col, n_nodes, n_rel, M, cap, gstart):
    """Returns dict(hdr [nb15,8] int32, mask [nb15] int32, loc [nb15,256] int64 (-1 = never written),
    src [nb15,cap,2] int64 (first n_src rows of live items valid))."""
    n_blocks = (n_nodes + M - 1) // M
    nb15 = n_blocks * LC_UNITS
    hdr = np.zeros((nb15, HDR), np.int32)
    mask = np.zeros(nb15, np.int32)
    loc = np.full((nb15, 256), -1, np.int64)
    src = np.zeros((nb15, cap, 2), np.int64)

    def finalize(b, u0, live, n_src, span):
        h = hdr[b * LC_UNITS + u0]
        h[:] = 0
        h[0], h[1], h[2] = len(live), n_src, span
        for j, u in enumerate(live):
            h[4 + j // 4] |= u << (8 * (j % 4))

    for b in range(n_blocks):
        for g in range(len(gstart) - 1):
            u_begin, u_end = gstart[g], gstart[g + 1]
            cur_u0, table, n_src, live = u_begin, {}, 0, []
            for u in range(u_begin, u_end):
                rows = []          # (i, deg, e0, key)
                m = 0
                for i in range(M):
                    n = b * M + i
                    deg, e0, key = 0, 0, -1
                    if n < n_nodes:
                        if u == 0:
                            deg, key = 1, n
                        else:
                            e0, e1 = csr_row(rowptr, n_rel, n, u - 1)
                            deg = e1 - e0
                            if deg == 1:
                                key = int(col[e0])
                    rows.append((i, deg, e0, key))
                    if deg > 0:
                        m |= 1 << (i >> 4)
                mask[b * LC_UNITS + u] = m
                if m == 0:
                    continue

                def winners(tbl):
                    seen, out = set(), []
                    for i, deg, e0, key in rows:
                        if deg >= 2:
                            out.append(i)
                        elif deg == 1 and key not in tbl and key not in seen:
                            seen.add(key)
                            out.append(i)
                    return out

                win = winners(table)
                if n_src + len(win) > cap:
                    finalize(b, cur_u0, live, n_src, u - cur_u0)
                    cur_u0, table, n_src, live = u, {}, 0, []
                    win = winners(table)
                ids = {i: n_src + k for k, i in enumerate(win)}
                item = b * LC_UNITS + cur_u0
                for i, deg, e0, key in rows:
                    if i in ids:
                        src[item, ids[i]] = (key, 0) if deg == 1 else (e0, deg)
                        if deg == 1:
                            table[key] = ids[i]
                for i, deg, e0, key in rows:
                    v = 0xFFFF
                    if deg == 1:
                        v = table[key]
                    elif deg >= 2:
                        v = ids[i]
                    loc[b * LC_UNITS + u, (i & 15) * 16 + (i >> 4)] = v
                n_src += len(win)
                live.append(u)
            finalize(b, cur_u0, live, n_src, u_end - cur_u0)
    return {"hdr": hdr, "mask": mask, "loc": loc, "src": src}


def plan_layout(n_nodes, M, cap):
    n = ((n_nodes + M - 1) // M) * LC_UNITS
    hdr = 0
    mask = hdr + n * HDR
    loc = mask + ((n + 3) & ~3)
    src = loc + n * 128
    return {"n": n, "hdr": hdr, "mask": mask, "loc": loc, "src": src, "total": src + n * cap * 2}


def split_device_plan(words, n_nodes, M, cap):
    """int32 words of a device plan -> the same dict layout as lc_plan_ref (loc as int64 of the uint16 values)."""
    L = plan_layout(n_nodes, M, cap)
    n = L["n"]
    w = np.asarray(words, np.int32)
    hdr = w[L["hdr"]:L["hdr"] + n * HDR].reshape(n, HDR)
    mask = w[L["mask"]:L["mask"] + n]
    loc = w[L["loc"]:L["loc"] + n * 128].view(np.uint16).reshape(n, 256).astype(np.int64)
    src = w[L["src"]:L["src"] + n * cap * 2].reshape(n, cap, 2).astype(np.int64)
    return {"hdr": hdr, "mask": mask, "loc": loc, "src": src}


def plan_edges(plan, col, n_nodes, M):
    """Expand a plan back into the multiset of (unit, destination row, source row) it will contract."""
    out = []
    nb15 = plan["hdr"].shape[0]
    for slot in range(nb15):
        n_live, n_src, span = (int(x) for x in plan["hdr"][slot, :3])
        if n_live <= 0:
            continue
        b, u0 = divmod(slot, LC_UNITS)
        units = [(int(plan["hdr"][slot, 4 + j // 4]) >> (8 * (j % 4))) & 0xff for j in range(n_live)]
        assert all(u0 <= u < u0 + span for u in units)
        for u in units:
            for i in range(M):
                v = int(plan["loc"][b * LC_UNITS + u, (i & 15) * 16 + (i >> 4)])
                if v == 0xFFFF or v < 0:
                    continue
                assert v < n_src
                a, d = (int(x) for x in plan["src"][slot, v])
                n = b * M + i
                if d == 0:
                    out.append((u, n, a))
                else:
                    out += [(u, n, int(col[a + k])) for k in range(d)]
    return sorted(out)


def csr_edges(rowptr, col, n_nodes, n_rel):
    out = [(0, n, n) for n in range(n_nodes)]
    for n in range(n_nodes):
        for r in range(n_rel):
            e0, e1 = csr_row(rowptr, n_rel, n, r)
            out += [(r + 1, n, int(col[e])) for e in range(e0, e1)]
    return sorted(out)


def csr_from_coo(us, vs, n_nodes):
    """Tile-major CSR by destination of lgcn_csr_build (numpy): rowptr [n_sub*n_rel*16+1], col sorted per row."""
    n_rel = len(us)
    n_sub = (n_nodes + 15) // 16
    nk = n_sub * n_rel * 16
    keys, cols = [], []
    for r, (u, v) in enumerate(zip(us, vs)):
        u = np.asarray(u, np.int64)
        keys.append(((u >> 4) * n_rel + r) * 16 + (u & 15))
        cols.append(np.asarray(v, np.int64))
    keys = np.concatenate(keys) if keys else np.zeros(0, np.int64)
    cols = np.concatenate(cols) if cols else np.zeros(0, np.int64)
    order = np.lexsort((cols, keys))
    rowptr = np.zeros(nk + 1, np.int64)
    np.add.at(rowptr, keys + 1, 1)
    return np.cumsum(rowptr).astype(np.int32), cols[order].astype(np.int32)

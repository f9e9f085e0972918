"""Executes a host-side plan (protocol_amd.Plan) with exact Python integers / numpy doubles.

TEST INFRASTRUCTURE: this walks the same slices, blocks, positions and partial-sum slots the HIP
kernels walk, so CPU tests can check the builder's layout and the multi-GPU exchange protocol
(own blocks + partial-sum tails + all-gather, or gather vectors + pack + all-to-all) without a GPU.
It is never used by the product.

Integer opinions are iterated the way the kernels iterate them ("raw form"): the vectors hold
s_j * inv(sum_j), an edge contributes opinion * entry, the plain scores of the share's own rows are kept
aside, and the step flagged final writes plain scores.  Field-valued opinions use normalised weights and
plain scores throughout.
"""
import numpy as np

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
INV_2_288 = pow(pow(2, 288, R), R - 2, R)
INV_2_576 = pow(pow(2, 576, R), R - 2, R)


class SegmentEmulator:
    def __init__(self, plan):
        self.p = plan
        info = plan.info
        ex = plan.exchange
        self.G, self.seg = info.n_seg, info.seg
        self.rows_cap, self.nd_cap = info.rows_cap, info.nd_cap
        self.stride = self.rows_cap + self.nd_cap
        self.gather = bool(ex.gather)
        self.vec_len = int(ex.vec_len)
        self.full_len = int(ex.full_len)
        self.base = int(ex.own_off)
        # where every share's partial sums of the dangling peers' score mass sit in this share's vectors
        self.dtail = [self.base + self.rows_cap if g == self.seg else int(ex.recv_off[g]) + int(ex.recv_count[g]) - self.nd_cap
                      for g in range(self.G)]
        self.raw = plan.value_u32 is not None
        self.inv_nm1 = plan.inv_nm1_w * INV_2_288 % R
        self.inv_nm1_f = 1.0 / (info.n_valid - 1)
        # factors of the own rows (position order on the host, whatever the layout of the vectors)
        first = self.seg * self.stride
        self.own_scale_fr = [s * INV_2_576 % R for s in plan.scale_fr[first:first + self.stride]]
        self.own_scale_f64 = plan.scale_f64[first:first + self.stride]
        self.own = {True: None, False: None}       # plain scores of the own rows, per variant (raw form)

    def _row_is_member(self, local):
        p = self.p
        q = ((local >> 6) * self.G + self.seg) * 64 + (local & 63)
        return q < p.n_active or (p.n_active_pad <= q < p.n_active_pad + self.p.info.n_dangling)

    def init(self, initial, field=True):
        """This share's own block of the initial vector (list of ints or float array of vec_len)."""
        x = [0] * self.vec_len if field else np.zeros(self.vec_len)
        own = [0] * self.rows_cap if field else np.zeros(self.rows_cap)
        for i in range(self.rows_cap):
            if self._row_is_member(i):
                own[i] = initial
                if not self.raw:
                    x[self.base + i] = initial
                elif field:
                    x[self.base + i] = initial * self.own_scale_fr[i] % R
                else:
                    x[self.base + i] = initial * self.own_scale_f64[i]
        self.own[field] = own
        d0 = self.p.n_dangling_local * initial
        x[self.base + self.rows_cap] = d0 % R if field else float(d0)
        return x

    def own_slice(self):
        return slice(self.base, self.base + self.stride)

    segment_slice = own_slice

    def pack(self, y, share):
        """What `share` cites of this share's own block of y, in its order, then the partial sums (gather vectors)."""
        rows = [int(r) for r in self.p.send[share]] + list(range(self.rows_cap, self.stride))
        return [y[self.base + r] for r in rows]

    def pack_self(self, y):
        """The entries of this share's own rows that its own rows cite, copied into their block of y (gather vectors)."""
        ex = self.p.exchange
        if not self.gather:
            return
        rows = [int(r) for r in self.p.need[self.seg]]
        assert len(rows) == int(ex.self_count)
        for j, r in enumerate(rows):
            y[int(ex.self_off) + j] = y[self.base + r]

    def recv_slice(self, share):
        ex = self.p.exchange
        return slice(int(ex.recv_off[share]), int(ex.recv_off[share]) + int(ex.recv_count[share]))

    def step(self, x, y, field=True, final=False):
        """One iteration over this share's rows: reads x, writes the own block of y."""
        p = self.p
        tails = [x[self.dtail[g] + k] for g in range(self.G) for k in range(self.nd_cap)]
        D = sum(tails) % R if field else float(np.sum(np.array(tails, dtype=np.float64)))
        own = self.own[field]
        changed = 0
        total = 0
        for (sb, se, flags, dslot) in p.work:
            dang = bool(flags & 1)
            dacc = 0
            for s in range(sb, se):
                off, ln, row0, nrows = p.slices[s]
                sh = (nrows >> 8) & 0xF            # log2(lanes per row)
                n_rows = nrows & 0xFF
                accs = [0] * 64
                for k in range(ln):
                    for lane in range(64):
                        e = (off + k) * 64 + lane
                        c = int(p.col[e])
                        c = (c >> 29) * self.stride + (c & 0x1FFFFFFF)   # (segment, local row) -> position; a gather index is itself
                        if field:
                            v = int(p.value_u32[e]) if self.raw else p.value_fr[e]
                            if v:
                                w = v if self.raw else v * p.scale_fr[c] % R * INV_2_576 % R
                                accs[lane] = (accs[lane] + w * x[c]) % R
                        else:
                            v = float(p.value_u32[e]) if self.raw else float(p.value_fr[e])
                            if v:
                                accs[lane] += (v if self.raw else v * p.scale_f64[c]) * x[c]
                group = 1 << sh
                rows = [(row0 + r, (sum(accs[r * group:(r + 1) * group]) % R) if field else float(sum(accs[r * group:(r + 1) * group])))
                        for r in range(n_rows)]
                for row, acc in rows:
                    old = own[row] if self.raw else x[self.base + row]
                    if field:
                        v = (acc + ((D - old) if dang else D) * self.inv_nm1) % R
                        changed += v != old
                        total = (total + v) % R
                    else:
                        v = acc + ((D - old) if dang else D) * self.inv_nm1_f
                        total += v
                    if self.raw:
                        own[row] = v
                        if final:
                            y[self.base + row] = v
                        elif field:
                            y[self.base + row] = v * self.own_scale_fr[row] % R
                        else:
                            y[self.base + row] = v * self.own_scale_f64[row]
                    else:
                        y[self.base + row] = v
                    if dang:
                        dacc = (dacc + v) % R if field else dacc + v
            if dang:
                y[self.base + self.rows_cap + dslot] = dacc
        return total, changed

    def unpermute(self, full, field=True):
        """full: a vector in position order (every share's own block, block g at g * stride)."""
        out = [0] * self.p.n if field else np.zeros(self.p.n)
        for i, pos in enumerate(self.p.pos_of_slot):
            if pos != 0xFFFFFFFF:
                out[i] = full[int(pos)]
        return out

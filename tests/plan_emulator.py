"""Executes a host-side plan (protocol_amd.Plan) with exact Python integers / numpy doubles.

TEST INFRASTRUCTURE: this walks the same slices, blocks, positions and partial-sum slots the HIP
kernels walk, so CPU tests can check the builder's layout and the multi-GPU exchange protocol
(segments + partial-sum tails + all-gather) without a GPU.  It is never used by the product.
"""
import numpy as np

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
INV_2_288 = pow(pow(2, 288, R), R - 2, R)
INV_2_576 = pow(pow(2, 576, R), R - 2, R)


class SegmentEmulator:
    def __init__(self, plan):
        self.p = plan
        info = plan.info
        self.G, self.seg = info.n_seg, info.seg
        self.rows_cap, self.nd_cap = info.rows_cap, info.nd_cap
        self.stride = self.rows_cap + self.nd_cap
        self.vec_len = self.stride * self.G
        self.base = self.seg * self.stride
        self.inv_nm1 = plan.inv_nm1_w * INV_2_288 % R
        self.inv_nm1_f = 1.0 / (info.n_valid - 1)

    def _row_is_member(self, local):
        p = self.p
        q = ((local >> 6) * self.G + self.seg) * 64 + (local & 63)
        return q < p.n_active or (p.n_active_pad <= q < p.n_active_pad + self.p.info.n_dangling)

    def init(self, initial, field=True):
        """This segment's part of the initial vector (list of ints or float array of vec_len)."""
        x = [0] * self.vec_len if field else np.zeros(self.vec_len)
        for i in range(self.rows_cap):
            if self._row_is_member(i):
                x[self.base + i] = initial
        d0 = self.p.n_dangling_local * initial
        x[self.base + self.rows_cap] = d0 % R if field else float(d0)
        return x

    def segment_slice(self):
        return slice(self.base, self.base + self.stride)

    def step(self, x, y, field=True):
        """One iteration over this segment's rows: reads all of x, writes this segment of y."""
        p = self.p
        tails = [x[g * self.stride + self.rows_cap + k] for g in range(self.G) for k in range(self.nd_cap)]
        D = sum(tails) % R if field else float(np.sum(np.array(tails, dtype=np.float64)))
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
                        c = (c >> 29) * self.stride + (c & 0x1FFFFFFF)   # (segment, local row) -> position
                        if field:
                            v = int(p.value_u32[e]) if p.value_u32 is not None else p.value_fr[e]
                            if v:
                                w = v * p.scale_fr[c] % R * INV_2_576 % R
                                accs[lane] = (accs[lane] + w * x[c]) % R
                        else:
                            v = float(p.value_u32[e]) if p.value_u32 is not None else float(p.value_fr[e])
                            if v:
                                accs[lane] += v * p.scale_f64[c] * x[c]
                group = 1 << sh
                rows = [(row0 + r, (sum(accs[r * group:(r + 1) * group]) % R) if field else float(sum(accs[r * group:(r + 1) * group])))
                        for r in range(n_rows)]
                for row, acc in rows:
                    old = x[self.base + row]
                    if field:
                        v = (acc + ((D - old) if dang else D) * self.inv_nm1) % R
                        changed += v != old
                        total = (total + v) % R
                    else:
                        v = acc + ((D - old) if dang else D) * self.inv_nm1_f
                        total += v
                    y[self.base + row] = v
                    if dang:
                        dacc = (dacc + v) % R if field else dacc + v
            if dang:
                y[self.base + self.rows_cap + dslot] = dacc
        return total, changed

    def unpermute(self, x, field=True):
        out = [0] * self.p.n if field else np.zeros(self.p.n)
        for i, pos in enumerate(self.p.pos_of_slot):
            if pos != 0xFFFFFFFF:
                out[i] = x[int(pos)]
        return out

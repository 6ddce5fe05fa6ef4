"""TEST-ONLY backend: lets host-side logic (dispatch, autograd wiring, halo exchange, partitioning)
run on CPU tensors by routing the arithmetic to the CPU oracle.  Registered by tests that need it
(never by the product: mi355x_graph has no CPU path and raises DGLError on CPU tensors)."""
import numpy as np
import torch

from mi355x_graph import sparse
from oracle import oracle as orc


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


class OracleBackend(object):
    name = "oracle"

    def degrees(self, csr):
        return (csr.indptr[1:] - csr.indptr[:-1]).to(csr.indptr.dtype)

    def inv_degrees(self, csr):
        d = (csr.indptr[1:] - csr.indptr[:-1]).clamp(min=1).to(torch.float32)
        return 1.0 / d

    def spmm(self, csr, op, reduce, U, E, u_len, e_len, out_len, u_off, e_off, src_scale, dst_scale, want_arg,
             accumulate_into=None):
        n = csr.num_rows
        L = orc.lib()
        Un = None if U is None else np.ascontiguousarray(_np(U).reshape(U.shape[0], -1), np.float32)
        if src_scale is not None and Un is not None:
            Un = Un * _np(src_scale)[:, None]
        En = None if E is None else np.ascontiguousarray(_np(E).reshape(E.shape[0], -1), np.float32)
        ip = np.ascontiguousarray(_np(csr.indptr), np.int32)
        ix = np.ascontiguousarray(_np(csr.indices), np.int32)
        ei = None if csr.eids is None else np.ascontiguousarray(_np(csr.eids), np.int32)

        def table(off, length):
            if off is not None:
                return np.ascontiguousarray(_np(off), np.int64)
            if length == out_len or length == 0:
                return None
            return (np.arange(out_len, dtype=np.int64) // (out_len // length))

        uo, eo = table(u_off, u_len), table(e_off, e_len)
        red = "sum" if reduce == "mean" else reduce
        out = np.empty((n, out_len), np.float32)
        au = np.full((n, out_len), -1, np.int32)
        ae = np.full((n, out_len), -1, np.int32)
        L.orc_spmm(orc._c64(n), orc._p(ip), orc._p(ix), orc._p(ei), orc._ci(orc.OPS[op]), orc._ci(orc.REDUCES[red]),
                   orc._p(Un), orc._p(En), orc._c64(u_len), orc._c64(e_len), orc._c64(out_len), orc._p(uo), orc._p(eo),
                   orc._p(out), orc._p(au), orc._p(ae))
        if reduce == "mean":
            out = out / np.maximum(np.diff(ip), 1).astype(np.float32)[:, None]
        if dst_scale is not None:
            out = out * _np(dst_scale)[:, None]
        idt = csr.indptr.dtype
        if accumulate_into is not None:
            accumulate_into.view(n, out_len).add_(torch.from_numpy(out))
            return accumulate_into.view(n, out_len), None, None
        return (torch.from_numpy(out), torch.from_numpy(au).to(idt) if want_arg and op != "copy_rhs" else None,
                torch.from_numpy(ae).to(idt) if want_arg and op != "copy_lhs" else None)

    def sddmm(self, gidx, op, L, R, lt, rt, l_len, r_len, out_len, reduce_size, l_off, r_off):
        src, dst = gidx.coo()
        nnz = gidx.num_edges()
        Ln = None if L is None else np.ascontiguousarray(_np(L).reshape(L.shape[0], -1), np.float32)
        Rn = None if R is None else np.ascontiguousarray(_np(R).reshape(R.shape[0], -1), np.float32)

        def table(off, length, full):
            if off is not None:
                return np.ascontiguousarray(_np(off), np.int64)
            if length == full or length == 0:
                return None
            return (np.arange(out_len, dtype=np.int64) // (full // length))

        full = out_len * reduce_size
        lo, ro = table(l_off, l_len, full), table(r_off, r_len, full)
        out = np.empty((nnz, out_len), np.float32)
        s32 = np.ascontiguousarray(_np(src), np.int32)
        d32 = np.ascontiguousarray(_np(dst), np.int32)
        orc.lib().orc_sddmm(orc._c64(nnz), orc._p(s32), orc._p(d32), orc._ci(orc.OPS[op]), orc._p(Ln), orc._p(Rn),
                            orc._ci(orc.TARGETS[lt]), orc._ci(orc.TARGETS[rt]), orc._c64(l_len), orc._c64(r_len),
                            orc._c64(out_len), orc._c64(reduce_size), orc._p(lo), orc._p(ro), orc._p(out))
        return torch.from_numpy(out)

    def edge_softmax_fwd(self, csr, z2d):
        return torch.from_numpy(orc.edge_softmax_fwd(_np(csr.indptr), _np(csr.eids), _np(z2d)))

    def edge_softmax_bwd(self, csr, a2d, da2d):
        return torch.from_numpy(orc.edge_softmax_bwd(_np(csr.indptr), _np(csr.eids), _np(a2d), _np(da2d)))

    def _gat_logits(self, csr, el2d, er2d, slope):
        rows = torch.repeat_interleave(torch.arange(csr.num_rows), (csr.indptr[1:] - csr.indptr[:-1]).long())
        t = el2d[csr.indices.long()] + er2d[rows]          # CSR-position order
        z = torch.where(t > 0, t, t * slope)
        if csr.eids is not None:                           # back to edge-id order
            out = torch.empty_like(z)
            out[csr.eids.long()] = z
            z, tt = out, torch.empty_like(t)
            tt[csr.eids.long()] = t
            t = tt
        return z.contiguous(), t

    def gat_attention_fwd(self, csr, el2d, er2d, slope):
        z, _ = self._gat_logits(csr, el2d, er2d, slope)
        return self.edge_softmax_fwd(csr, z)

    def gat_attention_bwd(self, csr, el2d, er2d, slope, a2d, da2d):
        _, t = self._gat_logits(csr, el2d, er2d, slope)
        dz = self.edge_softmax_bwd(csr, a2d, da2d)
        return dz * torch.where(t > 0, torch.ones_like(t), torch.full_like(t, slope))

    @staticmethod
    def head_dot_supported(H, F):
        return True

    def head_dot_fwd(self, feat3d, attn_a, attn_b):
        out_a = (feat3d.double() * attn_a.double()).sum(-1).float()
        out_b = (feat3d.double() * attn_b.double()).sum(-1).float() if attn_b is not None else None
        return out_a, out_b

    def head_dot_bwd(self, feat3d, attn_a, attn_b, d_a, d_b, need_feat_grad):
        x = feat3d.double()
        d_feat = d_a.double().unsqueeze(-1) * attn_a.double()
        g_a = (d_a.double().unsqueeze(-1) * x).sum(0).float()
        g_b = None
        if attn_b is not None:
            d_feat = d_feat + d_b.double().unsqueeze(-1) * attn_b.double()
            g_b = (d_b.double().unsqueeze(-1) * x).sum(0).float()
        return (d_feat.float() if need_feat_grad else None), g_a, g_b

    def column_pair_sums(self, a2d, b2d=None, shifted=False):
        a = a2d.double()
        if b2d is None:
            if shifted:
                a = a - a[0]
            return a.sum(0).float(), (a * a).sum(0).float()
        b = b2d.double()
        if shifted:
            b = b - b[0]
        return a.sum(0).float(), (a * b).sum(0).float()

    def column_affine(self, a2d, A, Cc, b2d=None, B=None):
        out = a2d * A + Cc
        return out if b2d is None else out + b2d * B

    COLUMN_SUM_MAX = 256
    XTY_MAX = (256, 1024)
    XTY_MIN_ROWS = 1 << 16

    def xty(self, a2d, b2d):
        return (a2d.double().t() @ b2d.double()).float()

    def column_sum(self, x2d):
        return x2d.double().sum(0).float()

    def segment_reduce(self, offsets, x2d, reduce, want_arg):
        return torch.from_numpy(orc.segment_reduce(_np(offsets), _np(x2d), reduce)), None

    # ---- halo rows as bitmaps + packed values (the contract of csrc/rowpack.hip, any bit order that pack and unpack share: here column order)
    @staticmethod
    def rows_pack_supported(x2d):
        return x2d.dim() == 2 and x2d.dtype == torch.float32 and x2d.shape[1] % 4 == 0 and 4 <= x2d.shape[1] <= 256

    @staticmethod
    def _bits(D):
        W = (D + 63) // 64
        return W, (torch.ones((), dtype=torch.int64) << (torch.arange(W * 64) % 64)).view(W, 64)

    def rows_pack_count(self, x2d, idx):
        x = x2d if idx is None else x2d[idx.long()]
        D = x.shape[1]
        W, bit = self._bits(D)
        nz = torch.zeros((x.shape[0], W * 64), dtype=torch.bool)
        nz[:, :D] = x != 0
        masks = (nz.view(-1, W, 64).to(torch.int64) * bit).sum(-1)
        return masks, nz.sum(1).to(torch.int32)

    def _flags(self, masks, D):
        W, bit = self._bits(D)
        return ((masks.view(-1, W, 1) & bit) != 0).view(masks.shape[0], W * 64)[:, :D]

    def rows_mask_count(self, masks, D):
        return self._flags(masks, D).sum(1).to(torch.int32)

    def rows_pack_values(self, x2d, idx, masks, offsets, total):
        x = x2d if idx is None else x2d[idx.long()]
        vals = x[self._flags(masks, x.shape[1])]
        assert vals.numel() == int(total)
        return vals.contiguous()

    def rows_unpack(self, masks, offsets, values, D, out=None):
        flags = self._flags(masks, D)
        dense = torch.zeros((masks.shape[0], D), dtype=torch.float32)
        dense[flags] = values
        if out is None:
            return dense
        out.copy_(dense)
        return out

    def gather_rows(self, x2d, idx):
        return x2d[idx.long()].contiguous()

    def scatter_add_rows(self, x2d, idx, rows2d):
        x2d.index_add_(0, idx.long(), rows2d)
        return x2d


def install():
    sparse.register_backend("cpu", OracleBackend())


def uninstall():
    sparse._BACKENDS.pop("cpu", None)

"""Dual sets: functionals + entity->dof maps; ``to_riesz`` assembles the
generalised Vandermonde / Riesz tensor on the device.

Mirrors FIAT/dual_set.py DualSet (:17-252).  to_riesz (:86-206): the expansion
set is tabulated ONCE at the sorted union of all functional points (HIP kernel)
and contracted with the weight tensor (fx_riesz_assemble)."""
import numpy
import torch

from . import runtime


class DualSet:
    def __init__(self, nodes, ref_el, entity_ids, entity_permutations=None):
        self.nodes = nodes
        self.ref_el = ref_el
        self.entity_ids = entity_ids
        self.entity_permutations = entity_permutations
        self.entity_closure_ids = {}
        for dim, entities in ref_el.sub_entities.items():
            self.entity_closure_ids[dim] = {}
            for e, subs in entities.items():
                ids = []
                for d, se in subs:
                    ids += self.entity_ids[d][se]
                self.entity_closure_ids[dim][e] = sorted(ids)

    def __iter__(self):
        return iter(self.nodes)

    def __len__(self):
        return len(self.nodes)

    def get_nodes(self):
        return self.nodes

    def get_entity_closure_ids(self):
        return self.entity_closure_ids

    def get_entity_ids(self):
        return self.entity_ids

    def get_entity_permutations(self):
        if self.entity_permutations is None:
            raise NotImplementedError("entity_permutations not yet implemented for %s" % type(self))
        return self.entity_permutations

    def get_reference_element(self):
        return self.ref_el

    def riesz_weights(self):
        """(points, W): sorted unique functional points and the dense weight tensor
        W[node, *target_shape, point] such that node(f) = sum W . f(points)."""
        if any(len(ell.deriv_dict) for ell in self.nodes):
            raise NotImplementedError("derivative functionals are not supported by fiat_amd yet")
        tshape = self.nodes[0].target_shape
        pts = sorted({pt for ell in self.nodes for pt in ell.pt_dict})
        index = {pt: j for j, pt in enumerate(pts)}
        W = numpy.zeros((len(self.nodes), *tshape, len(pts)))
        for i, ell in enumerate(self.nodes):
            for pt, wc_list in ell.pt_dict.items():
                j = index[pt]
                for w, c in wc_list:
                    W[(i, *c, j)] += w
        return numpy.array(pts, dtype=float), W

    def to_riesz(self, poly_set):
        """R[i, *shape, k] = node_i(e_shape phi_k) for the expansion set of poly_set."""
        es = poly_set.get_expansion_set()
        ed = poly_set.get_embedded_degree()
        nexp = es.get_num_members(ed)
        pts, W = self.riesz_weights()
        tshape = W.shape[1:-1]
        ctx = runtime.Context.get()
        if hasattr(es, "_device_set"):
            ev = es._device_set(ed).tabulate_batch(0, pts[None])[0, 0]      # (nexp, npts) on device
        else:  # 1-D Lagrange expansion set
            ev = es.device_line().tabulate_batch(0, pts.reshape(1, -1))[0, 0]
        Wd = torch.as_tensor(W.reshape(-1, W.shape[-1])).to(ctx.device)
        mat = runtime.riesz_assemble(Wd, ev, ctx)
        return mat.cpu().numpy().reshape((len(self.nodes), *tshape, nexp))

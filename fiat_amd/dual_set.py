"""Dual sets: functionals + entity->dof maps; ``to_riesz`` assembles the
generalised Vandermonde / Riesz tensor on the device.

Mirrors FIAT/dual_set.py DualSet (:17-252).  to_riesz (:86-206): the expansion
set is tabulated ONCE at the sorted union of all functional points (HIP kernel)
and contracted with the weight tensor (fx_riesz_assemble)."""
import numpy
import torch

from . import runtime


def merge_entities(nodes, ref_el, entity_ids, entity_permutations):
    """Dual set on a split cell -> dual set on the parent simplex (FIAT/dual_set.py:305-335): the dofs of
    all children of a parent entity are collected on that entity; point evaluations are re-ordered
    lexicographically by their barycentric coordinates on the parent (:291-302), other functionals keep
    their order."""
    parent = ref_el.get_parent()
    if parent is None:
        return nodes, ref_el, entity_ids, entity_permutations
    from . import functional
    from .macro import xy_to_bary
    children = ref_el.get_parent_to_children()
    parent_ids = {}
    if all(isinstance(node, functional.PointEvaluation) for node in nodes):
        merged = []
        for dim in sorted(children):
            parent_ids[dim] = {}
            for entity in sorted(children[dim]):
                first = len(merged)
                for cdim, centity in children[dim][entity]:
                    merged.extend(nodes[i] for i in entity_ids[cdim][centity])
                count = len(merged) - first
                if count > 1:
                    pts = [next(iter(node.pt_dict)) for node in merged[first:]]
                    order = numpy.lexsort(xy_to_bary(parent.get_vertices(), pts).T)
                    parent_ids[dim][entity] = [first + int(j) for j in order]
                else:
                    parent_ids[dim][entity] = list(range(first, first + count))
    else:
        merged = nodes
        for dim in sorted(children):
            parent_ids[dim] = {entity: [i for cdim, centity in children[dim][entity] for i in entity_ids[cdim][centity]]
                               for entity in sorted(children[dim])}
    return merged, parent, parent_ids, None


class DualSet:
    def __init__(self, nodes, ref_el, entity_ids, entity_permutations=None):
        nodes, ref_el, entity_ids, entity_permutations = merge_entities(nodes, ref_el, entity_ids, entity_permutations)
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

    def get_indices(self, restriction_domain, take_closure=True):
        """Dofs supported on a restriction domain: "interior" (the cell's own dofs), or all dofs on entities
        of dimension <= that of "vertex" / "edge" / "face" / "facet" / "ridge" (only that dimension when
        ``take_closure`` is false); FIAT/dual_set.py:208-252."""
        sd = self.ref_el.get_spatial_dimension()
        if restriction_domain == "interior":
            dims = [max(self.entity_ids)]
        else:
            try:
                dim = {"vertex": 0, "edge": 1, "face": 2, "facet": sd - 1, "ridge": sd - 2}[restriction_domain]
            except KeyError:
                raise RuntimeError("Invalid restriction domain")
            dims = range(0 if take_closure else dim, dim + 1)
        return [dof for d in dims for entity in sorted(self.entity_ids[d]) for dof in self.entity_ids[d][entity]]

    def riesz_weights(self):
        """(points, W): sorted unique functional points and the dense weight tensor
        W[node, *target_shape, point] such that the value part of node(f) = sum W . f(points);
        (None, None) when no functional evaluates function values."""
        tshape = self.nodes[0].target_shape
        pts = sorted({pt for ell in self.nodes for pt in ell.pt_dict})
        if not pts:
            return None, None
        index = {pt: j for j, pt in enumerate(pts)}
        W = numpy.zeros((len(self.nodes), *tshape, len(pts)))
        for i, ell in enumerate(self.nodes):
            for pt, wc_list in ell.pt_dict.items():
                j = index[pt]
                for w, c in wc_list:
                    W[(i, *c, j)] += w
        return numpy.array(pts, dtype=float), W

    def riesz_derivative_weights(self):
        """(points, order, D): the derivative part (FIAT/dual_set.py:175-205).  D[node, *target_shape, t, point]
        weights the t-th derivative table (mis() order over all orders <= ``order``; t = 0, the values, stays
        zero) at the sorted unique derivative points; (None, 0, None) without derivative functionals.  Any order the
        device tabulates (<= 8: orders above 2 through differentiation matrices)."""
        order = max(ell.max_deriv_order for ell in self.nodes)
        if order == 0:
            return None, 0, None
        from .polynomial_set import mis
        sd = self.ref_el.get_spatial_dimension()
        tshape = self.nodes[0].target_shape
        alphas = [a for k in range(order + 1) for a in mis(sd, k)]
        slot = {a: t for t, a in enumerate(alphas)}
        pts = sorted({pt for ell in self.nodes for pt in ell.deriv_dict})
        index = {pt: j for j, pt in enumerate(pts)}
        D = numpy.zeros((len(self.nodes), *tshape, len(alphas), len(pts)))
        for i, ell in enumerate(self.nodes):
            for pt, wac_list in ell.deriv_dict.items():
                j = index[pt]
                for w, alpha, c in wac_list:
                    D[(i, *c, slot[tuple(alpha)], j)] += w
        return numpy.array(pts, dtype=float), order, D

    def _expansion_tables(self, es, ed, pts, order):
        """Device tensor (ntab, nexp, npts): the expansion set and its derivatives <= order at pts."""
        if es.ref_el.get_spatial_dimension() == 0:     # the constant 1 on a point cell (expansions.ExpansionSet._tabulate)
            return torch.ones((1, 1, len(pts)), dtype=torch.float64, device=runtime.Context.get().device)
        if hasattr(es, "_device_set"):
            return es._device_set(ed).tabulate_batch(order, pts[None])[0]
        return es.device_line().tabulate_batch(order, pts.reshape(1, -1))[0]  # 1-D Lagrange expansion set

    def to_riesz(self, poly_set):
        """R[i, *shape, k] = node_i(e_shape phi_k) for the expansion set of poly_set."""
        es = poly_set.get_expansion_set()
        ed = poly_set.get_embedded_degree()
        nexp = es.get_num_members(ed)
        tshape = self.nodes[0].target_shape
        ctx = runtime.Context.get()
        mat = torch.zeros((len(self.nodes) * int(numpy.prod(tshape, dtype=int)), nexp), dtype=torch.float64, device=ctx.device)
        pts, W = self.riesz_weights()
        if pts is not None:
            ev = self._expansion_tables(es, ed, pts, 0)[0]                          # (nexp, npts) on device
            Wd = torch.as_tensor(W.reshape(-1, W.shape[-1])).to(ctx.device)
            mat += runtime.riesz_assemble(Wd, ev, ctx)
        dpts, order, D = self.riesz_derivative_weights()
        if dpts is not None:
            tabs = self._expansion_tables(es, ed, dpts, order)                      # (ntab, nexp, npts)
            ntab, _, npts = tabs.shape
            # the (table, point) pairs play the role of quadrature points
            ev = tabs.permute(1, 0, 2).reshape(nexp, ntab * npts).contiguous()
            Dd = torch.as_tensor(D.reshape(-1, ntab * npts)).to(ctx.device)
            mat += runtime.riesz_assemble(Dd, ev, ctx)
        return mat.cpu().numpy().reshape((len(self.nodes), *tshape, nexp))

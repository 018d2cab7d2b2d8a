// Host-side "plan" for one expansion set: the Dubiner recurrence flattened into
// a table of uniform three-term steps, plus the C0 (bubble) post-processing
// folded into a member-space matrix.  Device kernels only ever see the tables.
//
// Reference behaviour being reproduced: FIAT/expansions.py:140-267
// (dubiner_recurrence), :24-40 (jrc / integrated_jrc), :251-266 (per-codim
// normalisation) and :270-322 (C0_basis).  The reference rescales members in
// place after every codimension; here every member is kept in its *final*
// scaling and the ratios of the scale factors are folded into the step
// coefficients, so a step is always
//     m_dst = (A*fa - B*fb) * m_cur - C*fc * m_prv
// with (fa, fb, fc) the collapsed-coordinate factors of the step's codimension.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace fx {

struct Step {
    int dst, cur, prv, codim;  // member indices; prv < 0: two-term (first) step
    double A, B, C;
};

inline int member_index(int sd, const int* idx) {
    if (sd == 1) return idx[0];
    if (sd == 2) {
        int t = idx[0] + idx[1];
        return t * (t + 1) / 2 + idx[1];
    }
    int t = idx[0] + idx[1] + idx[2];
    int u = idx[1] + idx[2];
    return t * (t + 1) * (t + 2) / 6 + u * (u + 1) / 2 + idx[2];
}

inline int binom(int n, int k) {
    if (k < 0 || k > n) return 0;
    long long r = 1;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return (int)r;
}

// All m-tuples of non-negative integers summing to n, in the order of FIAT's mis() (polynomial_set.py:23-32): the
// first entry descends from n, the rest recursively -- (2,0,0) (1,1,0) (1,0,1) (0,2,0) (0,1,1) (0,0,2).
inline std::vector<std::vector<int>> multi_indices(int m, int n) {
    if (m == 1) return {{n}};
    std::vector<std::vector<int>> out;
    for (int i = 0; i <= n; ++i)
        for (const std::vector<int>& rest : multi_indices(m - 1, i)) {
            std::vector<int> a{n - i};
            a.insert(a.end(), rest.begin(), rest.end());
            out.push_back(a);
        }
    return out;
}

inline void jacobi_abc(double a, double b, int n, double& an, double& bn, double& cn) {
    double s = a + b;
    an = (2 * n + 1 + s) * (2 * n + 2 + s) / (2 * (n + 1) * (n + 1 + s));
    bn = s * (a - b) * (2 * n + 1 + s) / (2 * (n + 1) * (n + 1 + s) * (2 * n + s));
    cn = (n + a) * (n + b) * (2 * n + 2 + s) / ((n + 1) * (n + 1 + s) * (2 * n + s));
}

inline void integrated_jacobi_abc(double a, double b, int n, double& an, double& bn, double& cn) {
    if (n == 1) {
        an = (a + b + 2) / 2;
        bn = (a - 3 * b - 2) / 2;
        cn = 0.0;
    } else {
        jacobi_abc(a - 1, b + 1, n - 1, an, bn, cn);
    }
}

// sqrt(norm2) applied to the member with (length d = codim+1) index `idx`
// when codimension `codim` has been completed (expansions.py:251-266).
inline double level_norm(int variant, int codim, const int* idx) {
    int d = codim + 1;
    double norm2;
    if (variant != 0) {
        int shift = (variant == 2) ? 1 : 0;
        int p = idx[d - 1] + shift;
        int s = 0;
        for (int i = 0; i < d - 1; ++i) s += idx[i];
        int al = 2 * (s + d * shift) - 1;
        norm2 = (0.5 + d) / d;
        if (p > 0 && p + al > 0) norm2 *= double(p + al) * double(2 * p + al) / double(p);
    } else {
        int s = 0;
        for (int i = 0; i < d; ++i) s += idx[i];
        norm2 = double(2 * s + d) / double(d);
    }
    return std::sqrt(norm2);
}

// product of the normalisations applied at codimensions >= codim to the member
// whose index (zero padded to sd entries) is idx.
inline double tail_norm(int variant, int sd, int codim, const int* idx) {
    double w = 1.0;
    for (int c = codim; c < sd; ++c) w *= level_norm(variant, c, idx);
    return w;
}

struct Program {
    int sd = 0, n = 0, variant = 0, nexp = 0;
    double phi0 = 0.0;        // final value of member 0
    std::vector<Step> steps;  // in dependency order, chains contiguous
};

// enumerate prefixes (length codim) with sum < n in the reference's order
// (reference_element.py:64-76: last entry slowest).
inline void prefixes(int codim, int n, std::vector<std::vector<int>>& out) {
    out.clear();
    if (codim == 0) {
        out.push_back({});
    } else if (codim == 1) {
        for (int i = 0; i < n; ++i) out.push_back({i});
    } else {
        for (int last = 0; last < n; ++last)
            for (int first = 0; first < n - last; ++first) out.push_back({first, last});
    }
}

inline Program build_program(int sd, int n, int variant, double scale) {
    Program P;
    P.sd = sd;
    P.n = n;
    P.variant = variant;
    P.nexp = binom(n + sd, sd);
    if (variant == 1) scale = -scale;  // expansions.py:176-177
    int zero[3] = {0, 0, 0};
    if (n == 0) {  // expansions.py:193-194: returned before any normalisation
        P.phi0 = scale;
        return P;
    }
    P.phi0 = scale * tail_norm(variant, sd, 0, zero);
    const int beta = (variant == 2) ? 1 : 0;
    std::vector<std::vector<int>> subs;
    for (int codim = 0; codim < sd; ++codim) {
        prefixes(codim, n, subs);
        for (const auto& sub : subs) {
            int s = 0;
            for (int v : sub) s += v;
            double alpha, a, b, c;
            if (variant == 1) {
                alpha = 2 * s;
                a = b = -0.5;
            } else {
                alpha = 2 * s + codim;
                if (variant == 2) alpha += 1 + codim;
                a = 0.5 * (alpha + beta) + 1.0;
                b = 0.5 * (alpha - beta);
            }
            int len = n - s;  // members i = 0..len along this chain
            std::vector<int> id(len + 1);
            std::vector<double> lam(len + 1);
            for (int i = 0; i <= len; ++i) {
                int idx[3] = {0, 0, 0};
                for (int j = 0; j < codim; ++j) idx[j] = sub[j];
                idx[codim] = i;
                id[i] = member_index(sd, idx);
                lam[i] = tail_norm(variant, sd, codim, idx);
            }
            Step st;
            st.codim = codim;
            st.dst = id[1];
            st.cur = id[0];
            st.prv = -1;
            st.A = a * lam[1] / lam[0];
            st.B = b * lam[1] / lam[0];
            st.C = 0.0;
            P.steps.push_back(st);
            for (int i = 1; i < len; ++i) {
                if (variant == 1)
                    integrated_jacobi_abc(alpha, beta, i, a, b, c);
                else
                    jacobi_abc(alpha, beta, i, a, b, c);
                st.dst = id[i + 1];
                st.cur = id[i];
                st.prv = id[i - 1];
                st.A = a * lam[i + 1] / lam[i];
                st.B = b * lam[i + 1] / lam[i];
                st.C = c * lam[i + 1] / lam[i - 1];
                P.steps.push_back(st);
            }
        }
    }
    return P;
}

// Row-major nexp x nexp matrix T with C0_basis(phi) = T phi (expansions.py:270-322).
inline std::vector<double> c0_transform(int sd, int n) {
    int nexp = binom(n + sd, sd);
    std::vector<double> M((size_t)nexp * nexp, 0.0);
    for (int i = 0; i < nexp; ++i) M[(size_t)i * nexp + i] = 1.0;
    auto row_scale = [&](int r, double w) {
        for (int k = 0; k < nexp; ++k) M[(size_t)r * nexp + k] *= w;
    };
    auto row_sub = [&](int r, int src) {
        for (int k = 0; k < nexp; ++k) M[(size_t)r * nexp + k] -= M[(size_t)src * nexp + k];
    };
    auto ix2 = [&](int p, int q) { int t[3] = {p, q, 0}; return member_index(2, t); };
    auto ix3 = [&](int p, int q, int r) { int t[3] = {p, q, r}; return member_index(3, t); };
    row_scale(0, -1.0);
    for (int j = 1; j <= sd; ++j) row_sub(0, j);
    if (sd == 2) {
        for (int i = 2; i <= n; ++i) row_sub(ix2(0, i), ix2(1, i - 1));
    } else if (sd == 3) {
        for (int i = 2; i <= n; ++i) {
            for (int j = 0; j <= n - i; ++j) row_sub(ix3(0, i, j), ix3(1, i - 1, j));
            row_sub(ix3(0, 0, i), ix3(0, 1, i - 1));
            row_sub(ix3(0, 0, i), ix3(1, 0, i - 1));
        }
    }
    std::vector<int> dofs;
    for (int i = 0; i <= sd; ++i) dofs.push_back(i);
    if (sd == 1) {
        for (int i = 2; i <= n; ++i) dofs.push_back(i);
    } else if (sd == 2) {
        for (int i = 2; i <= n; ++i) dofs.push_back(ix2(1, i - 1));
        for (int i = 2; i <= n; ++i) dofs.push_back(ix2(0, i));
        for (int i = 2; i <= n; ++i) dofs.push_back(ix2(i, 0));
        for (int j = 1; j <= n; ++j)
            for (int i = 2; i <= n - j; ++i) dofs.push_back(ix2(i, j));
    } else {
        for (int i = 2; i <= n; ++i) dofs.push_back(ix3(0, 1, i - 1));
        for (int i = 2; i <= n; ++i) dofs.push_back(ix3(1, 0, i - 1));
        for (int i = 2; i <= n; ++i) dofs.push_back(ix3(1, i - 1, 0));
        for (int i = 2; i <= n; ++i) dofs.push_back(ix3(0, 0, i));
        for (int i = 2; i <= n; ++i) dofs.push_back(ix3(0, i, 0));
        for (int i = 2; i <= n; ++i) dofs.push_back(ix3(i, 0, 0));
        for (int j = 1; j <= n; ++j)
            for (int i = 2; i <= n - j; ++i) dofs.push_back(ix3(1, i - 1, j));
        for (int j = 1; j <= n; ++j)
            for (int i = 2; i <= n - j; ++i) dofs.push_back(ix3(0, i, j));
        for (int j = 1; j <= n; ++j)
            for (int i = 2; i <= n - j; ++i) dofs.push_back(ix3(i, 0, j));
        for (int j = 1; j <= n; ++j)
            for (int i = 2; i <= n - j; ++i) dofs.push_back(ix3(i, j, 0));
        for (int k = 1; k <= n; ++k)
            for (int j = 1; j <= n - k; ++j)
                for (int i = 2; i <= n - j - k; ++i) dofs.push_back(ix3(i, j, k));
    }
    std::vector<double> T((size_t)nexp * nexp);
    for (int r = 0; r < nexp; ++r)
        for (int k = 0; k < nexp; ++k) T[(size_t)r * nexp + k] = M[(size_t)dofs[r] * nexp + k];
    return T;
}

// MFMA A-operand fragments of the (rows x nexp) coefficient matrix for
// v_mfma_f64_16x16x4_f64: fragment (mt, ks), lane l holds
// C[16*mt + (l & 15)][4*ks + (l >> 4)], zero padded.
inline std::vector<double> pack_a_fragments(const std::vector<double>& C, int rows, int nexp) {
    int MT = (rows + 15) / 16, KS = (nexp + 3) / 4;
    std::vector<double> F((size_t)MT * KS * 64, 0.0);
    for (int mt = 0; mt < MT; ++mt)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l) {
                int m = 16 * mt + (l & 15), k = 4 * ks + (l >> 4);
                if (m < rows && k < nexp) F[((size_t)mt * KS + ks) * 64 + l] = C[(size_t)m * nexp + k];
            }
    return F;
}

// ---------------------------------------------------------------------------------
// Cooperative plan for large shapes (coop_kernel.hpp): the same steps as
// build_program, re-ordered depth first and split over NPROD producer waves.
// The (p,q) chains of the last codimension are the units of ownership: the owner
// of chain (p,q) publishes its head (p,q,0) and its members (p,q,r>0); the
// lower-codimension steps that lead to the head are executed by every wave that
// needs them (cheap) and published by the owner only.  Wave w's j-th published
// member occupies K slot 4*j + w, so every K-step takes exactly one member from
// every producer (zero rows pad the tail).
struct CoopEntry {
    int level;    // codimension of the step; -1: the constant member 0
    int seed;     // -2: continue the chain of `level`; -1: chain starts from the constant;
                  // 0/1: chain starts from the current member of that level
    int publish;  // K slot + 1 (1..4): store the result into that row of the K-step's slab; 0: keep in registers only
    int member;   // member index produced (for the K permutation), -1 for a zero pad
    double A, B, C;
};

struct CoopPlan {
    int sd = 0, n = 0, nexp = 0, KS = 0;
    double phi0 = 0.0;
    std::vector<CoopEntry> entries[4];
    std::vector<int> kstart[4];  // entries of K-step j: [kstart[j], kstart[j+1])
    std::vector<int> kperm;      // K slot (4*j + w) -> member, -1: none
};

inline CoopPlan build_coop_plan(const Program& P) {
    CoopPlan C;
    C.sd = P.sd;
    C.n = P.n;
    C.nexp = P.nexp;
    C.phi0 = P.phi0;
    const int sd = P.sd, n = P.n;
    // step that produces each member
    std::vector<int> step_of(P.nexp, -1);
    for (size_t i = 0; i < P.steps.size(); ++i) step_of[P.steps[i].dst] = (int)i;
    auto mid = [&](int p, int q, int r) { int t[3] = {p, q, r}; return member_index(sd, t); };
    // ownership units: chains of the last codimension, keyed by their prefix
    struct Unit { int p, q, size; };
    std::vector<Unit> units;
    if (sd == 1) {
        units.push_back({0, 0, n + 1});
    } else if (sd == 2) {
        for (int p = 0; p <= n; ++p) units.push_back({p, 0, n - p + 1});
    } else {
        for (int p = 0; p <= n; ++p)
            for (int q = 0; q <= n - p; ++q) units.push_back({p, q, n - p - q + 1});
    }
    std::vector<int> order(units.size());
    for (size_t i = 0; i < units.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return units[a].size > units[b].size; });
    int load[4] = {0, 0, 0, 0};
    std::vector<int> owner(units.size(), 0);
    for (int u : order) {
        int w = 0;
        for (int k = 1; k < 4; ++k)
            if (load[k] < load[w]) w = k;
        owner[u] = w;
        load[w] += units[u].size;
    }
    auto owner_of = [&](int p, int q) {
        for (size_t i = 0; i < units.size(); ++i)
            if (units[i].p == p && units[i].q == q) return owner[i];
        return 0;
    };
    for (int w = 0; w < 4; ++w) {
        std::vector<CoopEntry>& E = C.entries[w];
        int cur_p = 0, cur_q = 0;  // members currently held by levels 0 and 1
        auto emit_step = [&](int member, int publish) {
            const Step& st = P.steps[step_of[member]];
            CoopEntry e;
            e.level = st.codim;
            e.publish = publish;
            e.member = member;
            e.A = st.A;
            e.B = st.B;
            e.C = st.C;
            if (st.prv >= 0) {
                e.seed = -2;
            } else {  // chain start: where does the seed member live?
                int seedm = st.cur;
                if (seedm == 0) {
                    e.seed = -1;
                } else {
                    e.seed = P.steps[step_of[seedm]].codim;  // level that produced the seed
                }
            }
            E.push_back(e);
        };
        for (size_t ui = 0; ui < units.size(); ++ui) {  // units are in (p, q) lexicographic order
            if (owner[ui] != w) continue;
            const int p = units[ui].p, q = units[ui].q;
            // level 0: advance to (p,0,0); published by the owner of chain (p,0) only
            // (sd == 1 has a single unit with p == 0, so this loop never runs there)
            while (cur_p < p) {
                ++cur_p;
                emit_step(mid(cur_p, 0, 0), owner_of(cur_p, 0) == w ? 1 : 0);
                cur_q = 0;
            }
            if (p == 0 && q == 0) {  // the constant member heads chain (0,0)
                CoopEntry e;
                e.level = -1;
                e.seed = -1;
                e.publish = 1;
                e.member = 0;
                e.A = e.B = e.C = 0.0;
                E.push_back(e);
            }
            if (sd == 1) {
                for (int i = 1; i <= n; ++i) emit_step(mid(i, 0, 0), 1);
                continue;
            }
            if (sd == 2) {  // unit = chain p along q (codim 1)
                for (int i = 1; i <= n - p; ++i) emit_step(mid(p, i, 0), 1);
                continue;
            }
            // sd == 3: level 1 advances to (p,q,0), then the codim-2 chain
            while (cur_q < q) {
                ++cur_q;
                emit_step(mid(p, cur_q, 0), owner_of(p, cur_q) == w ? 1 : 0);
            }
            for (int r = 1; r <= n - p - q; ++r) emit_step(mid(p, q, r), 1);
        }
    }
    // K slots.  The producers meet at a barrier after every K-step, and a K-step of the contraction
    // takes four published members from whichever producers supply them: all publishes are ordered
    // by the time their producer reaches them (entries executed so far, steps of the LDS-resident
    // levels 0/1 weigh 1.5), K-step j takes publishes 4j .. 4j+3 and ends at the time T_j of the
    // last of them; every producer executes in K-step j whatever it can finish by T_j, published
    // or not, so that all four stay busy between barriers.  A fixed "one member per producer per K-step" rule made every K-step as slow
    // as the producer that had a run of unpublished prefix steps in it (DG P6 tet: the sum over
    // K-steps of the longest producer was 45 entries per request for 29 per producer).
    // CoopEntry::publish = K slot + 1 (0: not published).
    auto cost = [](const CoopEntry& e) { return (e.level == 0 || e.level == 1) ? 1.5 : 1.0; };
    struct Pub { double t; int w, idx; };
    std::vector<Pub> pubs;
    std::vector<double> tcum[4];  // time at which producer w has finished entry i
    for (int w = 0; w < 4; ++w) {
        double t = 0.0;
        for (size_t i = 0; i < C.entries[w].size(); ++i) {
            t += cost(C.entries[w][i]);
            tcum[w].push_back(t);
            if (C.entries[w][i].publish) pubs.push_back({t, w, (int)i});
        }
    }
    std::stable_sort(pubs.begin(), pubs.end(), [](const Pub& a, const Pub& b) { return a.t < b.t; });
    const int total_pubs = (int)pubs.size();
    const int KS = (total_pubs + 3) / 4;
    C.KS = KS;
    C.kperm.assign((size_t)4 * KS, -1);
    std::vector<int> group[4];  // K-step of every published entry (-1: not published)
    for (int w = 0; w < 4; ++w) {
        C.kstart[w].assign(1, 0);
        group[w].assign(C.entries[w].size(), -1);
    }
    for (int k = 0; k < total_pubs; ++k) group[pubs[k].w][pubs[k].idx] = k / 4;
    for (int j = 0; j < KS; ++j) {
        double T = 0.0;  // the K-step ends when its last member is published
        for (int slot = 0; slot < 4 && 4 * j + slot < total_pubs; ++slot) {
            const Pub& pb = pubs[(size_t)4 * j + slot];
            CoopEntry& e = C.entries[pb.w][pb.idx];
            C.kperm[(size_t)4 * j + slot] = e.member;
            e.publish = slot + 1;
            T = std::max(T, pb.t);
        }
        for (int w = 0; w < 4; ++w) {
            // everything this producer can do by then, published or not -- but a member of a
            // later K-step (ties in time) must wait: it would land in this K-step's slab
            size_t i = (size_t)C.kstart[w].back();
            while (i < C.entries[w].size() && tcum[w][i] <= T + 1e-9 && group[w][i] <= j) ++i;
            if (j == KS - 1) i = C.entries[w].size();
            C.kstart[w].push_back((int)i);
        }
    }
    const int used_last = total_pubs - 4 * (KS - 1);
    // zero rows for the unused slots of the last K-step (0 x garbage could be NaN): producer 0
    for (int slot = used_last; slot < 4; ++slot) {
        CoopEntry e;
        e.level = -1;
        e.seed = -2;  // marks a zero row
        e.publish = slot + 1;
        e.member = -1;
        e.A = e.B = e.C = 0.0;
        C.entries[0].push_back(e);
        C.kstart[0][KS] = (int)C.entries[0].size();
    }
    return C;
}

// Fragments for the shape-specialised kernel (simplex_fixed.hpp): full 16-row
// tiles first ([MT16][KS][64]), then 4-row blocks for v_mfma_f64_4x4x4_4b
// ([M4][KS][64]; lane l holds C[base + (l & 3)][4*ks + (l >> 4)], the block bits
// (l >> 2) & 3 see the same A).  A remainder of 13..15 rows uses a padded 16-row tile.
inline std::vector<double> pack_a_fragments_split(const std::vector<double>& C, int rows, int nexp,
                                                  const std::vector<int>* kperm = nullptr) {
    // kperm (optional): K slot -> member index (production order of the recurrence)
    auto col = [&](int k) { return kperm ? (*kperm)[k] : k; };  // -1: no member in this K slot
    int rem = rows % 16;
    bool split = rem != 0 && rem <= 12;
    int MT16 = split ? rows / 16 : (rows + 15) / 16;
    int M4 = split ? (rem + 3) / 4 : 0;
    // with a permutation the K extent is the permutation's length (may include empty slots)
    int nk = kperm ? (int)kperm->size() : nexp;
    int KS = (nk + 3) / 4;
    std::vector<double> F((size_t)(MT16 + M4) * KS * 64, 0.0);
    for (int mt = 0; mt < MT16; ++mt)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l) {
                int m = 16 * mt + (l & 15), k = 4 * ks + (l >> 4);
                if (m < rows && k < nk && col(k) >= 0) F[((size_t)mt * KS + ks) * 64 + l] = C[(size_t)m * nexp + col(k)];
            }
    for (int m4 = 0; m4 < M4; ++m4)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l) {
                int m = 16 * MT16 + 4 * m4 + (l & 3), k = 4 * ks + (l >> 4);
                if (m < rows && k < nk && col(k) >= 0) F[((size_t)(MT16 + m4) * KS + ks) * 64 + l] = C[(size_t)m * nexp + col(k)];
            }
    return F;
}

}  // namespace fx

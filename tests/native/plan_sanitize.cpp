// Host-only exercise of the planning code (fiat_amd/csrc/plan.hpp) for the AddressSanitizer / UBSan build of the
// CPU test-suite (sanitizers on the host library; GPU sanitizers are not available on this pool).
// Walks every table the C ABI builds on the host: recurrence programs, C0 transforms, cooperative schedules, A-fragment
// packings and multi-index lists for all (sd, degree, variant) the kernels are registered for, and checks the
// invariants the device code relies on (indices in range, every member produced exactly once, K slots a permutation).
#include <cstdio>
#include <cstdlib>
#include <set>

#include "../../fiat_amd/csrc/plan.hpp"

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "%s:%d: %s failed (sd %d n %d variant %d)\n", __FILE__, __LINE__, #cond, sd, n, variant); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main() {
    int programs = 0;
    for (int sd = 1; sd <= 3; ++sd)
        for (int variant = 0; variant <= 2; ++variant)
            for (int n = (variant == 1 ? 1 : 0); n <= (sd == 3 ? 8 : 10); ++n) {
                const fx::Program P = fx::build_program(sd, n, variant, 0.75);
                const int nexp = fx::binom(n + sd, sd);
                CHECK(P.nexp == nexp);
                CHECK((int)P.steps.size() == nexp - 1);
                std::set<int> made{0};
                for (const fx::Step& s : P.steps) {
                    CHECK(s.dst > 0 && s.dst < nexp && s.cur >= 0 && s.cur < nexp && s.prv >= -1 && s.prv < nexp);
                    CHECK(s.codim >= 0 && s.codim < sd);
                    CHECK(made.count(s.cur) == 1 && (s.prv < 0 || made.count(s.prv) == 1));   // operands exist already
                    CHECK(made.insert(s.dst).second);                                           // produced once
                    CHECK(std::isfinite(s.A) && std::isfinite(s.B) && std::isfinite(s.C));
                }
                CHECK((int)made.size() == nexp);
                ++programs;
                if (n >= 1) {
                    const std::vector<double> T = fx::c0_transform(sd, n);
                    CHECK((int)T.size() == nexp * nexp);
                    const fx::CoopPlan C = fx::build_coop_plan(P);
                    CHECK(C.KS == (nexp + 3) / 4);
                    std::set<int> slots;
                    for (int k = 0; k < 4 * C.KS; ++k)
                        if (C.kperm[k] >= 0) CHECK(C.kperm[k] < nexp && slots.insert(C.kperm[k]).second);
                    CHECK((int)slots.size() == nexp);
                    for (int w = 0; w < 4; ++w) CHECK((int)C.kstart[w].size() == C.KS + 1);
                }
                // coefficient matrices of awkward row counts through every packing
                for (int rows : {1, 3, 16, 20, 45, 60, 84}) {
                    std::vector<double> Cm((size_t)rows * nexp);
                    for (size_t i = 0; i < Cm.size(); ++i) Cm[i] = (double)(i % 97) - 48.0;
                    const std::vector<double> F = fx::pack_a_fragments(Cm, rows, nexp);
                    const std::vector<double> F2 = fx::pack_a_fragments_split(Cm, rows, nexp);
                    std::vector<int> perm(nexp);
                    for (int i = 0; i < nexp; ++i) perm[i] = i == 0 ? 0 : P.steps[i - 1].dst;
                    const std::vector<double> F3 = fx::pack_a_fragments_split(Cm, rows, nexp, &perm);
                    CHECK(!F.empty() && F2.size() == F3.size());
                    double s2 = 0.0, s3 = 0.0;      // a permutation of K keeps the multiset of entries
                    for (double v : F2) s2 += v;
                    for (double v : F3) s3 += v;
                    CHECK(std::fabs(s2 - s3) < 1e-9 * (1.0 + std::fabs(s2)));
                }
            }
    for (int m = 1; m <= 3; ++m)
        for (int k = 0; k <= 8; ++k) {
            const int sd = m, n = k, variant = 0;
            const auto A = fx::multi_indices(m, k);
            CHECK((int)A.size() == fx::binom(k + m - 1, m - 1));
            for (const auto& a : A) {
                int s = 0;
                for (int x : a) s += x;
                CHECK((int)a.size() == m && s == k);
            }
        }
    std::printf("plan_sanitize: %d programs ok\n", programs);
    return 0;
}

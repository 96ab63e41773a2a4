// Host-side plan builder under AddressSanitizer + UBSan (the GPU pool runs no sanitizers: the CPU build is where they can run).
// Builds plans over a sweep of shapes, flattens their tables, builds the row maps, and cross-checks a few invariants.
#include <cstdio>
#include <cstdlib>

#include "../../cliora_amd/csrc/plan.hpp"

using namespace cliora;

static int check(const Plan& p) {
    const int L = p.L, C = p.C;
    if (C != L * (L + 1) / 2) return 1;
    if ((int)p.pair_a_in.size() != p.P_in || (int)p.pair_a_out.size() != p.P_out) return 2;
    for (int v : p.pair_a_in) if (v < 0 || v >= C) return 3;
    for (int v : p.pair_b_out) if (v < 0 || v >= C) return 4;
    if ((int)p.level_geom.size() != 2 * L * PLEVEL_INTS) return 5;
    for (int pass = 0; pass < 2; ++pass)
        for (int lv = 0; lv < L; ++lv) {
            const int32_t* e = p.level_geom.data() + ((size_t)pass * L + lv) * PLEVEL_INTS;
            const int N = pass ? L - lv - 1 : lv;
            if (e[0] != L - lv || e[1] != N) return 6;
            if (e[5] < 1 || e[5] > 8 || e[6] < 1 || e[6] > HP_PARTS || e[7] < 1) return 7;
            const int G = (p.B * e[0] + 15) / 16;
            if (e[7] != (G + e[5] - 1) / e[5] * e[6]) return 8;          // ntask = groups of TG cell tiles x SP parts
        }
    for (int r = 0; r < N_ROLES; ++r) {
        const UseList& u = p.uses[r];
        if ((int)u.off.size() != C + 1 || u.off[C] != (int)u.row.size()) return 9;
    }
    return 0;
}

int main() {
    int n = 0;
    for (int arch = 0; arch < 2; ++arch)
        for (int L : {1, 2, 3, 7, 10, 20, 33, 64})
            for (int D : {1, 16, 50, 64, 400, 512})
                for (int B : {1, 3, 64})
                    for (int share = 0; share < 2; ++share) {
                        if ((long long)B * L * L * L > 3000000) continue;
                        Plan p;
                        const std::string e = build_plan(p, B, L, D, share, 1, 0, arch);
                        if (!e.empty()) { printf("build_plan(%d,%d,%d): %s\n", B, L, D, e.c_str()); return 10; }
                        const int rc = check(p);
                        if (rc) { printf("check %d failed for B %d L %d D %d share %d arch %d\n", rc, B, L, D, share, arch); return rc; }
                        std::vector<int32_t> flat = flatten_tables(p);
                        if (flat.empty()) return 11;
                        if (p.dev.level_geom + p.level_geom.size() > flat.size()) return 12;
                        build_row_maps(p);
                        if (p.arow.size() != (size_t)(p.R_in + p.R_out)) return 13;
                        if (!find_table(p, "use_row_outb") || find_table(p, "nope")) return 14;
                        ++n;
                    }
    Plan bad;
    if (build_plan(bad, 1, 65, 16, 1, 1, 0, 0).empty()) return 20;      // L > 64 is refused
    if (build_plan(bad, 1, 4, 16, 1, 1, 3, 1).empty()) return 21;       // TreeLSTM with image regions is refused
    printf("plans ok: %d\n", n);
    return 0;
}

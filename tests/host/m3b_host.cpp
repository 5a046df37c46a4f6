// tests/host/m3b_host.cpp -- the serial MACS 3D core of the product (tap-net_amd/csrc/tap_macs3_big.h: what
// k_macs3d_big_step runs per thread, and the control skeleton of the wave-per-container kernel) compiled for the HOST and
// stepped against the CPU oracle on random blocks: positions, stable flags, height-map, valid / empty counters per step, and
// that both sides raise on the same step.  Built and run by tests/test_macs3_big_host_cpu.py (no GPU needed).
#define M3B_HD
#include "tap_macs3_big.h"
#include "tap_oracle.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static int stab_host(int bx, int by, m3b_u64 eq) {
    uint8_t m[64]; for (int i = 0; i < bx; ++i) for (int j = 0; j < by; ++j) m[i*by+j] = (eq >> (i*8+j)) & 1;
    return orc_is_stable_3d_mask(bx, by, m);
}
int main(int argc, char **argv) {
    int W = atoi(argv[1]), L = atoi(argv[2]), H = atoi(argv[3]), n = atoi(argv[4]), eps = atoi(argv[5]), flags = atoi(argv[6]), hi = atoi(argv[7]), hz = atoi(argv[8]);
    unsigned seed = argc > 9 ? atoi(argv[9]) : 1;
    srand(seed);
    orc_desc d = {3, W, L, H, n, ORC_MACS, flags, ORC_R_CPS, ORC_FEAT_FULL};
    int cells = W*L, HW = (H+63)/64, cap = 128 + 8*n;
    long steps = 0, bad = 0, errs = 0;
    for (int ep = 0; ep < eps; ++ep) {
        orc_env *e = orc_env_new(&d);
        std::vector<int32_t> hm(cells, 0), pos(3*n, 0), blk(3*n, 0), lev(cells), slots(cells), lvh(n+2), lvr(n+2);
        std::vector<m3b_u64> occ((size_t)cells*HW, 0);
        std::vector<M3BEms> ems(cap);
        int cnt[4] = {0,0,0,0};
        for (int t = 0; t < n; ++t) {
            int32_t b[3] = {1 + rand() % (hi-1), 1 + rand() % (hi-1), 1 + rand() % (hz-1)};
            if (b[0] > W) b[0] = W; if (b[1] > L) b[1] = L;
            int rc = orc_env_add_block(e, b, nullptr);
            M3BState s = {W, L, H, HW, flags, cap, cnt[3], hm.data(), occ.data(), pos.data(), blk.data(), 1, ems.data(), lev.data(), slots.data(), lvh.data(), lvr.data()};
            int err = 0;
            M3BResult r = m3b_place(s, cnt, err, b[0], b[1], b[2], stab_host);
            pos[t*3] = r.x; pos[t*3+1] = r.y; pos[t*3+2] = r.z;
            blk[t*3] = b[0] | (r.placed << 16); blk[t*3+1] = b[1]; blk[t*3+2] = b[2];
            cnt[3] += 1;
            ++steps;
            if (rc != 0 || err) { ++errs; if ((rc != 0) != (err != 0)) { printf("ERR MISMATCH ep %d t %d rc %d err %d\n", ep, t, rc, err); ++bad; } break; }
            const int32_t *op = orc_env_positions(e), *oh = orc_env_heightmap(e);
            bool ok = op[t*3] == r.x && op[t*3+1] == r.y && op[t*3+2] == r.z && orc_env_stable(e)[t] == r.stab && memcmp(oh, hm.data(), 4*cells) == 0
                      && orc_env_valid(e) == cnt[0] && orc_env_empty(e) == cnt[1];
            if (!ok) { printf("MISMATCH ep %d t %d block %d %d %d: oracle (%d %d %d st %d) mine (%d %d %d st %d placed %d) valid %ld/%d empty %ld/%d\n", ep, t, b[0], b[1], b[2], op[t*3], op[t*3+1], op[t*3+2], orc_env_stable(e)[t], r.x, r.y, r.z, r.stab, r.placed, (long)orc_env_valid(e), cnt[0], (long)orc_env_empty(e), cnt[1]); ++bad; break; }
        }
        orc_env_free(e);
    }
    printf("W %d L %d H %d n %d flags %d: %ld steps, %ld mismatching episodes, %ld episodes ended by an error\n", W, L, H, n, flags, steps, bad, errs);
    return bad != 0;
}

// Host-only check of the NT GEMMs' tile order (video-tokenizer_amd/csrc/vt_common.h: vt_tile_of, vt_auto_col_block): for every tile grid up to
// 70 x 26 and every block width, the list must visit each (tile row, tile column) exactly once, and the automatic width must be a legal one.
// Built with hipcc for the host (no kernel is launched): tests/test_host_cpu.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../video-tokenizer_amd/csrc/vt_common.h"

int main() {
    long checked = 0;
    for (int tm_n = 1; tm_n <= 70; ++tm_n)
        for (int tn_n = 1; tn_n <= 26; ++tn_n)
            for (int W = 0; W <= 27; ++W) {
                const int w = W < tn_n ? W : 0;                       // what the launchers pass: widths >= the tile columns mean row-major
                std::vector<char> seen((size_t)tm_n * tn_n, 0);
                for (int sid = 0; sid < tm_n * tn_n; ++sid) {
                    int tm = -1, tn = -1;
                    vt_tile_of(sid, tm_n, tn_n, w, tm, tn);
                    if (tm < 0 || tm >= tm_n || tn < 0 || tn >= tn_n || seen[(size_t)tm * tn_n + tn]++) {
                        printf("FAIL grid %d x %d W %d: entry %d -> (%d, %d)\n", tm_n, tn_n, w, sid, tm, tn);
                        return 1;
                    }
                }
                ++checked;
            }
    for (int tn_n = 1; tn_n <= 64; ++tn_n)
        for (int fl = 1; fl <= 64; ++fl) {
            const int w = vt_auto_col_block(tn_n, fl);
            if (w < 0 || w >= (tn_n > 1 ? tn_n : 2)) { printf("FAIL auto width %d for %d tile columns, %d in flight\n", w, tn_n, fl); return 1; }
        }
    // the choices the design text quotes
    if (vt_auto_col_block(12, 32) != 6 || vt_auto_col_block(16, 32) != 6 || vt_auto_col_block(4, 32) != 0 || vt_auto_col_block(6, 9) != 3) {
        printf("FAIL quoted widths: %d %d %d %d\n", vt_auto_col_block(12, 32), vt_auto_col_block(16, 32), vt_auto_col_block(4, 32), vt_auto_col_block(6, 9));
        return 1;
    }
    printf("OK %ld orders\n", checked);
    return 0;
}

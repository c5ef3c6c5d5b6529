/* TEST INFRASTRUCTURE ONLY -- C restatement of the pieces of the U2PL hot path
 * whose CPU arithmetic must be reproduced bit-for-bit and needs fmaf().
 *
 * oracle_bilinear_ac: torch CPU F.interpolate(mode="bilinear",
 *   align_corners=True) as used at train_semi.py:320-322,345-350,355,372-374
 *   (SURVEY.md Appendix A, Q8):
 *     scale = f32((in-1)/(out-1));  src = f32(dst)*scale;  i0 = floor(src);
 *     i1 = min(i0+1, in-1);  l1 = src - i0;  l0 = 1 - l1;
 *     out = fma(ly0, fma(lx0, v00, lx1*v01), ly1 * fma(lx0, v10, lx1*v11))
 * Pinned against torch CPU in tests/test_oracle_golden.py (bit-exact).
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared restate.c -o _build/liboracle.so -lm
 */
#include <math.h>
#include <stddef.h>

static float ac_scale(long in, long out) {
    return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f;
}

void oracle_bilinear_ac(const float* x, float* y, long planes, long h, long w,
                        long oh, long ow) {
    const float sy = ac_scale(h, oh), sx = ac_scale(w, ow);
    for (long p = 0; p < planes; ++p) {
        const float* xp = x + (size_t)p * h * w;
        float* yp = y + (size_t)p * oh * ow;
        for (long oy = 0; oy < oh; ++oy) {
            const float fy = (float)oy * sy;
            long y0 = (long)fy;
            long y1 = y0 + (y0 < h - 1 ? 1 : 0);
            const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
            for (long ox = 0; ox < ow; ++ox) {
                const float fx = (float)ox * sx;
                long x0 = (long)fx;
                long x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
                const float v00 = xp[y0 * w + x0], v01 = xp[y0 * w + x1];
                const float v10 = xp[y1 * w + x0], v11 = xp[y1 * w + x1];
                const float top = fmaf(lx0, v00, lx1 * v01);
                const float bot = fmaf(lx0, v10, lx1 * v11);
                yp[oy * ow + ox] = fmaf(ly0, top, ly1 * bot);
            }
        }
    }
}

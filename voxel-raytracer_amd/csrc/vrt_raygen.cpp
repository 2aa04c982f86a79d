// vrt_raygen.cpp -- the part of ray generation (shaders/raytracing.comp:624-641) that does not depend on the pixel as a whole,
// made once per projection on the host. Pure arithmetic: no device, no context (also linked into the test-support library).
#include <cmath>
#include <cstring>
#include <vector>

namespace vrt_internal {

// ---- ray generation, the part that does not depend on the pixel as a whole ---------------------------------------------
// comp:626-634: u = px / W * 2 - 1, v likewise, view = invProjection * (u, v, -1, 1), view /= view.w. For the inverse of
// any perspective or orthographic projection x depends on u alone, y on v alone, z and w on neither, so the W + H + 1
// distinct values are computed here once per (matrix, W, H) -- by the SAME float operations in the same order, which is
// what makes them the same bits (this file is compiled with -ffp-contract=off like the kernels; x86-64 float arithmetic is
// IEEE binary32) -- and the kernels read them (View::gen_x/gen_y/gen_z) instead of making two integer->float conversions,
// five divisions and a 4x4 product per pixel. A zero term's SIGN can depend on the other coordinate ((+0) * v): each value
// is therefore evaluated with the other coordinate at +1 and at -1 and must come out the same bits, otherwise the table
// is refused and the kernels run the shader's own prologue. Returns false when the matrix is not of that shape.
bool build_ray_table(const float *m, int W, int H, std::vector<float> &tab, float &z_out) {
    static const int kZeros[10] = {4, 8, 12, 1, 9, 13, 2, 6, 3, 7};   // column-major m[c * 4 + r]
    for (int k : kZeros)
        if (!(m[k] == 0.0f)) return false;
    const auto row = [&](int r, float u, float v) {   // mat_vec() of vrt_common.hip.h with (x, y, z, w) = (u, v, -1, 1)
        return (m[0 * 4 + r] * u + m[1 * 4 + r] * v) + (m[2 * 4 + r] * -1.0f + m[3 * 4 + r] * 1.0f);
    };
    const auto same = [](float a, float b) { return std::memcmp(&a, &b, sizeof a) == 0; };
    const float w = row(3, 1.0f, 1.0f), z = row(2, 1.0f, 1.0f);
    for (int k = 1; k < 4; ++k) {
        const float u = (k & 1) ? -1.0f : 1.0f, v = (k & 2) ? -1.0f : 1.0f;
        if (!same(row(3, u, v), w) || !same(row(2, u, v), z)) return false;
    }
    if (!(w == w) || !(z == z)) return false;
    const bool divide = fabsf(w) > 1e-6f;
    tab.resize((size_t)W + (size_t)H);
    double hi = 0.0;
    for (int px = 0; px < W; ++px) {
        const float u = ((float)px / (float)W) * 2.0f - 1.0f;
        const float x = row(0, u, 1.0f);
        if (!same(x, row(0, u, -1.0f))) return false;
        const float q = divide ? x / w : x;
        if (!(fabs((double)q) <= 1.0995e12)) return false;   // 2^40; also refuses NaN
        hi = fabs((double)q) > hi ? fabs((double)q) : hi;
        tab[(size_t)px] = q;
    }
    for (int py = 0; py < H; ++py) {
        const float v = ((float)py / (float)H) * 2.0f - 1.0f;
        const float y = row(1, 1.0f, v);
        if (!same(y, row(1, -1.0f, v))) return false;
        const float q = divide ? y / w : y;
        if (!(fabs((double)q) <= 1.0995e12)) return false;
        tab[(size_t)W + (size_t)py] = q;
    }
    z_out = divide ? z / w : z;
    // range of the first normalisation, dot = (x^2 + y^2) + z^2 >= z^2: inside [2^-80, 2^82], its root inside [2^-40, 2^41]
    const double az = fabs((double)z_out);
    return az >= 9.0949e-13 && az <= 1.0995e12;   // 2^-40 .. 2^40
}

// The second normalisation takes |invView3x3 * d| for a unit d: between the matrix' smallest and largest singular value.
// With F2 the squared Frobenius norm, sigma_max <= sqrt(F2) and sigma_min = |det| / (sigma_1 sigma_2) >= 2 |det| / F2.
// True when that keeps the squared length inside [2^-62, 2^42] with room for the rounding of the product.
bool view_matrix_in_range(const float *m) {
    double a[3][3], f2 = 0.0;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) { a[r][c] = (double)m[c * 4 + r]; f2 += a[r][c] * a[r][c]; }
    if (!(f2 >= 9.0949e-13 && f2 <= 1.0995e12)) return false;      // 2^-40 .. 2^40, refuses NaN
    const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                       a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    return det * det >= 2.3842e-7 * f2 * f2 * f2;                   // sigma_min >= 2^-10 sqrt(F2)  <=  det^2 >= 2^-22 F2^3
}

}  // namespace vrt_internal

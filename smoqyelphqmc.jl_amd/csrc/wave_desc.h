// Lane programs of the one-wavefront kernels (fdm_wave_kernel in kernels_fdm_wave.hip, lanczos_wave_kernel in kernels_kpm.hip): the
// compile-time part of a program — which own position pairs with which under each colour, in this lane or in a neighbouring one, and
// which bond slot holds the bond — plus the helpers both kernels unroll with.  The run-time part (site ids, padded-bond indices, partner
// lanes) is FdmWave::tab, found and verified on the host (kernels_fdm_wave.hip, fdm_wave_program).
#pragma once
#include <type_traits>
#include <utility>

namespace smoqy {
namespace wave_desc {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- lane programs (compile-time part; the host side below holds the same tables: GroupDesc) --------------------------------
// pp(c, p): position of the partner of own position p under colour c (in this lane, or in the lane of row rr(c, p) of the table when
// rr >= 0); bs(c, p): the lane's bond slot holding that bond's (cosh, sinh).
struct RingD {
    static constexpr int S = 4, NCOL = 2, NB = 5, NR = 2, KIND = 1;
    static constexpr bool REMOTE0 = false;
    __host__ __device__ static constexpr int pp(int c, int p) { constexpr int t[2][4] = {{1, 0, 3, 2}, {3, 2, 1, 0}}; return t[c][p]; }
    __host__ __device__ static constexpr int rr(int c, int p) { constexpr int t[2][4] = {{-1, -1, -1, -1}, {1, -1, -1, 0}}; return t[c][p]; }
    __host__ __device__ static constexpr int bs(int c, int p) { constexpr int t[2][4] = {{0, 0, 1, 1}, {4, 2, 2, 3}}; return t[c][p]; }
};
struct PlaqD {
    static constexpr int S = 4, NCOL = 4, NB = 12, NR = 8, KIND = 2;
    static constexpr bool REMOTE0 = true;
    __host__ __device__ static constexpr int pp(int c, int p) { constexpr int t[4][4] = {{1, 0, 3, 2}, {1, 0, 3, 2}, {3, 2, 1, 0}, {3, 2, 1, 0}}; return t[c][p]; }
    __host__ __device__ static constexpr int rr(int c, int p) { constexpr int t[4][4] = {{0, 1, 2, 3}, {-1, -1, -1, -1}, {-1, -1, -1, -1}, {4, 5, 6, 7}}; return t[c][p]; }
    __host__ __device__ static constexpr int bs(int c, int p) { constexpr int t[4][4] = {{4, 5, 6, 7}, {0, 0, 1, 1}, {3, 2, 2, 3}, {8, 9, 10, 11}}; return t[c][p]; }
};
// 2 x 2 unit cells: cell k = dx + 2 dy holds A at position 2k and B at 2k + 1; colour 0 = A–B of a cell, colour 1 = A(x) – B(x − 1),
// colour 2 = A(y) – B(y − 1); rows of the lane table: 0 left, 1 right, 2 below, 3 above
struct HoneyD {
    static constexpr int S = 8, NCOL = 3, NB = 16, NR = 4, KIND = 3;
    static constexpr bool REMOTE0 = false;
    __host__ __device__ static constexpr int pp(int c, int p)
    {
        constexpr int t[3][8] = {{1, 0, 3, 2, 5, 4, 7, 6}, {3, 2, 1, 0, 7, 6, 5, 4}, {5, 4, 7, 6, 1, 0, 3, 2}};
        return t[c][p];
    }
    __host__ __device__ static constexpr int rr(int c, int p)
    {
        constexpr int t[3][8] = {{-1, -1, -1, -1, -1, -1, -1, -1}, {0, -1, -1, 1, 0, -1, -1, 1}, {2, -1, 2, -1, -1, 3, -1, 3}};
        return t[c][p];
    }
    __host__ __device__ static constexpr int bs(int c, int p)
    {
        constexpr int t[3][8] = {{0, 0, 1, 1, 2, 2, 3, 3}, {6, 4, 4, 8, 7, 5, 5, 9}, {12, 10, 13, 11, 10, 14, 11, 15}};
        return t[c][p];
    }
};

template <int CTRL>
__device__ __forceinline__ double wave_rot(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

}  // namespace wave_desc
}  // namespace smoqy

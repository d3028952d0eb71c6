// soar_m3.h -- the 3x3 matrix helpers of the per-Gaussian stages (preprocess_point.h, geom_bwd_point.h): products evaluated left to
// right as written, no FMA contraction, which is the operation order of the reference's matrix library (glm) under an IEEE evaluation.
#pragma once
#include "soar_common.h"

namespace soar {

namespace {

// 3x3 matrix addressed [column][row]; product evaluates r[c][r] = a[0][r]*b[c][0] + a[1][r]*b[c][1] + a[2][r]*b[c][2]
// left to right, which is the operation order the reference's matrix library uses.
struct M3 {
    float e[3][3];
};
__device__ __forceinline__ M3 m3mul(const M3 &a, const M3 &b)
{
#pragma clang fp contract(off)
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 3; k++) r.e[c][k] = a.e[0][k] * b.e[c][0] + a.e[1][k] * b.e[c][1] + a.e[2][k] * b.e[c][2];
    return r;
}
__device__ __forceinline__ M3 m3t(const M3 &a)
{
#pragma clang fp contract(off)
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 3; k++) r.e[c][k] = a.e[k][c];
    return r;
}

}  // namespace

}  // namespace soar

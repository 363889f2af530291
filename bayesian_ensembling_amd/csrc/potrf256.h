// potrf256.h - host interface of the register-resident 256 x 256 diagonal-block Cholesky (potrf256.hip; fp64 only)
#pragma once
#include <hip/hip_runtime.h>

#include "prims.h"

namespace bayens {
namespace potrf256n {
// Ablk: the 256 x 256 diagonal block of every member (lower triangle read; on return its lower factor, zeros above the
// diagonal inside the four diagonal 64-blocks); inv: the four inverted 64 x 64 diagonal blocks [4][64][64] per member
// (member stride inv_stride); a non-positive pivot j records info[b] = pivot_base + j + 1 if info[b] was 0.
void potrf_diag256(hipStream_t stream, int nb, Mat Ablk, double* inv, long inv_stride, int pivot_base, int* info);
// diagnostic: the cycle-counter stamps [8 waves][16] of workgroup 0 of the last launch
void potrf256_stamps(unsigned long long* out128);
}  // namespace potrf256n
}  // namespace bayens

// streaming 1x1 convolution (pw_stream.hip): plan + launch, shared with the composite executors (blocks.hip)
#pragma once
#include <hip/hip_runtime.h>
namespace hs {
struct PwArgs {
    const char* A;
    const char* W;
    char* D;
    unsigned long long a_bytes, w_bytes;
    int lda, ldd;
    int M, N, K;
    int nblocks;               // ceil(M / 64)
    float* stats;              // partial rows [(rows)][N][3] = (count, mean, M2) of D's columns, or NULL
};

struct PwPlan {
    int ok;            // 1: the streaming kernel takes this shape
    int bn, sm, grid, stat_rows, lds;
};
PwPlan pw_stream_plan(long long M, int N, int K, bool want_stats);
int pw_stream_run(const PwPlan& pl, const PwArgs& a, hipStream_t s);
}  // namespace hs

// yp_epilogue.h -- the reference decode (DetectNet.predict_transform, detect_net.py:152-183) of ONE
// (cell, anchor) record whose raw values are in registers: the hook a kernel that PRODUCES the
// head calls in its epilogue, so that the raw head never makes the round trip through memory
// between the network's last layer (detect_net.py:67-68, Dense(8450, relu)) and the decode.
// Included by yolopost.hip inside its anonymous namespace (AnchorTable, yp_math.h).
//
// k_decode_grid_ref, the kernel behind the drop-in predict_transform, evaluates its records with
// this function too, so a producer that calls the hook gets the bits that kernel would have
// produced from the stored head.  k_dense_relu_decode_ref below is such a producer: the dense
// layer as a plain dot product per output (a demonstration of the hook, not a tuned GEMM -- the
// model itself is outside this library) with the decode fused behind it.
#pragma once

// t: the 5 + C raw values of record n of an image (n = (row * W + col) * A + anchor).  C = CS when
// CS > 0 (the class exponentials are then evaluated once and kept in registers), else c_arg.
// Writes the record's centre, size, objectness and C class probabilities.
template <int CS>
__device__ __forceinline__ void yp_decode_record_ref(const float *t, int n, int H, int W, int A, int c_arg,
                                                     const AnchorTable &at, float2 &ce, float2 &sz,
                                                     float &obj, float *probs)
{
    const int C = CS ? CS : c_arg;
    const int a = n % A, k = n / A;
    const float gx = (float)(k % H), gy = (float)(k / H);   // detect_net.py:155-162
    ce.x = (yp_sigmoid(t[0]) + gx) / (float)H;              // :166-168
    ce.y = (yp_sigmoid(t[1]) + gy) / (float)W;
    sz.x = (yp_exp(t[2]) * at.anc[2 * a + 0]) / (float)H;   // :173-177
    sz.y = (yp_exp(t[3]) * at.anc[2 * a + 1]) / (float)W;
    obj = yp_sigmoid(t[4]);                                 // :171
    float m = t[5];                                         // softmax, :180
    for (int j = 1; j < C; ++j) m = t[5 + j] > m ? t[5 + j] : m;
    float s = 0.f;
    if (CS) {
#pragma unroll
        for (int j = 0; j < CS; ++j) probs[j] = yp_exp(t[5 + j] - m);
#pragma unroll
        for (int j = 0; j < CS; ++j) s = s + probs[j];
#pragma unroll
        for (int j = 0; j < CS; ++j) probs[j] = probs[j] / s;
    } else {
        for (int j = 0; j < C; ++j) s = s + yp_exp(t[5 + j] - m);
        for (int j = 0; j < C; ++j) probs[j] = yp_exp(t[5 + j] - m) / s;
    }
}

// head[b, n, v] = relu(bias[n, v] + sum_f features[b, f] * weight[f, n, v]) (detect_net.py:67-68),
// accumulated in the order f = 0, 1, ... with separate multiply and add, then decoded from the
// registers it was accumulated in.  Thread <-> record (b, n).  raw != null: the head is stored as
// well (what a caller needs to check the fused result against the two-step path).
template <int CS>
__global__ void k_dense_relu_decode_ref(const float *features, const float *weight, const float *bias, int F,
                                        int H, int W, int A, AnchorTable at, float *raw, float *center,
                                        float *wh, float *obj, float *cls, long long total)
{
    constexpr int V = 5 + CS;
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const int N = H * W * A;
    const long long b = o / N;
    const int n = (int)(o - b * N);
    float t[V];
#pragma unroll
    for (int v = 0; v < V; ++v) t[v] = bias[(size_t)n * V + v];
    const float *x = features + b * F;
    for (int f = 0; f < F; ++f) {
        const float xf = x[f];
        const float *wr = weight + ((size_t)f * N + n) * V;
#pragma unroll
        for (int v = 0; v < V; ++v) t[v] = t[v] + xf * wr[v];
    }
#pragma unroll
    for (int v = 0; v < V; ++v) t[v] = t[v] > 0.0f ? t[v] : 0.0f;
    if (raw) {
#pragma unroll
        for (int v = 0; v < V; ++v) raw[o * V + v] = t[v];
    }
    float2 ce, sz;
    float ob, probs[CS];
    yp_decode_record_ref<CS>(t, n, H, W, A, CS, at, ce, sz, ob, probs);
    reinterpret_cast<float2 *>(center)[o] = ce;
    reinterpret_cast<float2 *>(wh)[o] = sz;
    obj[o] = ob;
#pragma unroll
    for (int j = 0; j < CS; ++j) cls[o * CS + j] = probs[j];
}

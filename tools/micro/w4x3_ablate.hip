// Timing harness for conv_wino4_x3 with pieces of its K loop removed (AV2X_W4X3_ABLATE, see csrc/conv_wino4_x3.hip): which of the
// side streams (B fragments from L2, the split, the transforms, the gathers, the LDS traffic) the MFMA steps wait for.  Results of
// the ablated builds are numerically meaningless; only the durations are read.  Build + run: tools/micro/w4x3_ablate.sh
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../airv2x_perception_amd/csrc/conv_wino4_x3.hip"

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4, h = argc > 2 ? atoi(argv[2]) : 100, w = argc > 3 ? atoi(argv[3]) : 352, cin = argc > 4 ? atoi(argv[4]) : 256,
              cout = argc > 5 ? atoi(argv[5]) : cin;
    const size_t in_e = (size_t)n * h * w * cin, out_e = (size_t)n * h * w * cout;
    float *in, *out, *wp, *shift;
    void* u;
    hipMalloc(&in, in_e * 4); hipMalloc(&out, out_e * 4); hipMalloc(&wp, 9ull * cin * cout * 4); hipMalloc(&shift, cout * 4);
    hipMalloc(&u, av2x_wino4_x3_weight_bytes(cin, cout));
    std::vector<float> hin(in_e), hw(9ull * cin * cout);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hin) v = rnd();
    for (auto& v : hw) v = rnd() * 0.05f;
    hipMemcpy(in, hin.data(), in_e * 4, hipMemcpyHostToDevice);
    hipMemcpy(wp, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemset(shift, 0, cout * 4);
    if (av2x_wino4_x3_pack_weights(wp, cin, cout, u, nullptr)) { printf("pack failed: %s\n", av2x_last_error()); return 1; }
    av2x_conv_desc d{};
    d.n = n; d.h = h; d.w = w; d.cin = cin; d.in_ctot = cin; d.in_coff = 0; d.ho = h; d.wo = w; d.cout = cout; d.coutp = cout;
    d.out_ctot = cout; d.out_coff = 0; d.ks = 3; d.stride = 1; d.pad = 1; d.relu = 1; d.mode = AV2X_CONV; d.up = 1;
    d.tile = 0x60000400 | (32 << 16) | 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // both forms of the kernel (AV2X_W4X3_PP is read per launch): 0 = four waves, one per SIMD; 1 = eight waves, ping-pong (round 6).
    // The K-loop ablation macros apply to the four-wave form only.
    for (int pp = 0; pp < 2; ++pp) {
        setenv("AV2X_W4X3_PP", pp ? "1" : "0", 1);
        if (pp && AV2X_W4X3_ABLATE) break;
        for (int i = 0; i < 3; ++i)
            if (av2x::wino4_x3_dispatch(&d, in, u, nullptr, shift, nullptr, out, nullptr)) { printf("launch failed: %s\n", av2x_last_error()); return 1; }
        hipDeviceSynchronize();
        const int iters = 20;
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) av2x::wino4_x3_dispatch(&d, in, u, nullptr, shift, nullptr, out, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * n * h * w * cout * 9 * cin;
        printf("ablate=%d pp=%d n=%d %dx%d %d->%d: %.1f us  (%.1f TF bf16 executed)\n", AV2X_W4X3_ABLATE, pp, n, h, w, cin, cout, ms * 1e3 / iters,
               flops * 0.25 * 6 / (ms * 1e-3 / iters) / 1e12);
    }
    return 0;
}

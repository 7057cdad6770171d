// PCIe-inclusive rate of the host-buffer entry point opusgpu_decode_packets (include/opusgpu.h): packets in host memory in,
// PCM in host memory out, driven from C++ (tools/host_path_rate.py drives the same call through Python / ctypes).
// build:  g++ -O2 -std=c++17 tools/host_path_rate.cpp -Iinclude -Lesp32-opus-player_amd -lopusgpu -Wl,-rpath,'$ORIGIN/../esp32-opus-player_amd' -o build_exp/host_path_rate
// usage (GPU box):  build_exp/host_path_rate [streams] [steps]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "opusgpu.h"

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 65536, steps = argc > 2 ? atoi(argv[2]) : 6, L = 160;
    opusgpu_ctx *ctx;
    if (opusgpu_ctx_create(0, &ctx) != OPUSGPU_OK) { fprintf(stderr, "no device\n"); return 1; }
    if (opusgpu_streams_alloc(ctx, n, 2) != OPUSGPU_OK) { fprintf(stderr, "%s\n", opusgpu_last_error(ctx)); return 1; }
    std::vector<uint32_t> x(n);
    for (int s = 0; s < n; s++) x[s] = 0x9E3779B9u ^ (uint32_t)s;
    std::vector<uint8_t> bytes((size_t)n * (L + 1));
    std::vector<const uint8_t *> ptr(n);
    std::vector<int32_t> ids(n), lens(n, L + 1), res(n);
    std::vector<int16_t> pcm((size_t)n * 960 * 2);
    double total = 0;
    for (int f = 0; f < steps; f++) {
        for (int s = 0; s < n; s++) { // CELT-FB stereo, per-stream LCG payload
            uint8_t *p = &bytes[(size_t)s * (L + 1)];
            p[0] = 0xFC;
            for (int i = 0; i < L; i++) { x[s] = x[s] * 1664525u + 1013904223u; p[1 + i] = (uint8_t)(x[s] >> 24); }
            ptr[s] = p;
            ids[s] = s;
        }
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = opusgpu_decode_packets(ctx, n, ids.data(), ptr.data(), lens.data(), pcm.data(), 1, res.data());
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rc != OPUSGPU_OK) { fprintf(stderr, "decode_packets: %d %s\n", rc, opusgpu_last_error(ctx)); return 1; }
        for (int s = 0; s < n; s++) if (res[s] != 960) { fprintf(stderr, "frame %d of step %d: %d\n", s, f, res[s]); return 1; }
        printf("step %d: %.2f ms\n", f, ms);
        if (f >= 2) total += ms; // the first two steps grow the staging buffers
    }
    const double avg = total / (steps - 2);
    printf("host path: %d streams, %.2f ms per step = %.2f M frames/s (%.1f MB of PCM out, %.1f MB of packets in per step)\n", n, avg,
           n / avg / 1e3, n * 3840.0 / 1e6, n * (L + 1.0) / 1e6);
    opusgpu_ctx_destroy(ctx);
    return 0;
}

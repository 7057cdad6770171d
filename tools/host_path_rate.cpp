// PCIe-inclusive rate of the host-buffer entry point opusgpu_decode_packets (include/opusgpu.h): packets in host memory in,
// PCM in host memory out, driven from C++ (tools/host_path_rate.py drives the same call through Python / ctypes).
// build:  g++ -O2 -std=c++17 -pthread tools/host_path_rate.cpp -Iinclude -Lesp32-opus-player_amd -lopusgpu -Wl,-rpath,'$ORIGIN/../esp32-opus-player_amd' -o build_exp/host_path_rate
// usage (GPU box):  build_exp/host_path_rate [streams] [steps] [pinned] [contexts]
//   pinned = 1: the caller's PCM buffer is page-locked (opusgpu_host_register), so the PCM travels straight into it.
//   contexts = C > 1: the streams are split over C contexts, each driven by its own host thread: one context's framing and kernels
//   run under the other contexts' PCM copies (a call is synchronous; calls on different contexts are independent).
// Prints a checksum of the last step's PCM: equal for every variant.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "opusgpu.h"

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 65536, steps = argc > 2 ? atoi(argv[2]) : 6, L = 160;
    const int pinned = argc > 3 ? atoi(argv[3]) : 0, C = argc > 4 ? atoi(argv[4]) : 1;
    if (C < 1 || C > 8 || n % C || steps < 3) { fprintf(stderr, "bad arguments\n"); return 1; }
    const int per = n / C;
    std::vector<opusgpu_ctx *> ctx(C);
    for (int c = 0; c < C; c++) {
        if (opusgpu_ctx_create(0, &ctx[c]) != OPUSGPU_OK) { fprintf(stderr, "no device\n"); return 1; }
        if (opusgpu_streams_alloc(ctx[c], per, 2) != OPUSGPU_OK) { fprintf(stderr, "%s\n", opusgpu_last_error(ctx[c])); return 1; }
    }
    // every step's packets up front (CELT-FB stereo, per-stream LCG payload): the threads only decode
    std::vector<uint32_t> x(n);
    for (int s = 0; s < n; s++) x[s] = 0x9E3779B9u ^ (uint32_t)s;
    std::vector<std::vector<uint8_t>> bytes(steps, std::vector<uint8_t>((size_t)n * (L + 1)));
    for (int f = 0; f < steps; f++)
        for (int s = 0; s < n; s++) {
            uint8_t *p = &bytes[f][(size_t)s * (L + 1)];
            p[0] = 0xFC;
            for (int i = 0; i < L; i++) { x[s] = x[s] * 1664525u + 1013904223u; p[1 + i] = (uint8_t)(x[s] >> 24); }
        }
    std::vector<int32_t> ids(n), lens(n, L + 1), res(n);
    for (int s = 0; s < n; s++) ids[s] = s % per;
    std::vector<int16_t> pcm((size_t)n * 960 * 2);
    if (pinned && opusgpu_host_register(ctx[0], pcm.data(), pcm.size() * sizeof(int16_t)) != OPUSGPU_OK) {
        fprintf(stderr, "host_register: %s\n", opusgpu_last_error(ctx[0]));
        return 1;
    }
    std::vector<int> bad(C, 0);
    auto run = [&](int c, int f0, int f1) {
        std::vector<const uint8_t *> ptr(per);
        for (int f = f0; f < f1; f++) {
            for (int s = 0; s < per; s++) ptr[s] = &bytes[f][(size_t)(c * per + s) * (L + 1)];
            const int rc = opusgpu_decode_packets(ctx[c], per, ids.data() + c * per, ptr.data(), lens.data(), pcm.data() + (size_t)c * per * 1920, 1,
                                                  res.data() + c * per);
            if (rc != OPUSGPU_OK) { fprintf(stderr, "decode_packets: %d %s\n", rc, opusgpu_last_error(ctx[c])); bad[c] = 1; return; }
            for (int s = 0; s < per; s++) if (res[c * per + s] != 960) { bad[c] = 1; return; }
        }
    };
    auto all = [&](int f0, int f1) {
        std::vector<std::thread> th;
        for (int c = 0; c < C; c++) th.emplace_back(run, c, f0, f1);
        for (auto &t : th) t.join();
    };
    all(0, 2); // the first two steps grow the staging buffers
    const auto t0 = std::chrono::steady_clock::now();
    all(2, steps);
    const double avg = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (steps - 2);
    for (int c = 0; c < C; c++) if (bad[c]) { fprintf(stderr, "a step failed\n"); return 1; }
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < pcm.size(); i++) h = (h ^ (uint16_t)pcm[i]) * 1099511628211ull;
    if (pinned) opusgpu_host_unregister(ctx[0], pcm.data());
    printf("host path%s, %d context(s), PCM fnv1a %016llx: %d streams, %.2f ms per step = %.2f M frames/s (%.1f MB of PCM out, %.1f MB of packets in per step)\n",
           pinned ? " (page-locked PCM buffer)" : "", C, (unsigned long long)h, n, avg, n / avg / 1e3, n * 3840.0 / 1e6, n * (L + 1.0) / 1e6);
    for (int c = 0; c < C; c++) opusgpu_ctx_destroy(ctx[c]);
    return 0;
}

// tools/host_threads_c.cpp -- one-item host-pointer Encaps + Decaps pairs from T host threads of a C++ program (no interpreter
// lock in the way, unlike tools/host_threads.py): pairs per second against T, with the library's engine lanes (default) -- run it
// a second time with MLKEM_HOST_LANES=0 for the one-engine behaviour.  Every pair is checked (K == K', status 0).
// build: g++ -O2 -std=c++17 -pthread -Iinclude -o tools/host_threads_c.bin tools/host_threads_c.cpp -Lcrystals-kyber_amd -lmlkem_amd -Wl,-rpath,'$ORIGIN/../crystals-kyber_amd'
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "mlkem_batch.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int R = 1500;
    unsigned ekl, dkl, cl;
    mlkem_sizes(768, &ekl, &dkl, &cl);
    const char* lanes = getenv("MLKEM_HOST_LANES");
    for (int T : {1, 2, 4, 8, 16, 32, 64}) {
        std::atomic<int> ready{0}, bad{0};
        std::atomic<bool> go{false};
        std::vector<double> secs(T);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                std::vector<uint8_t> d(32, (uint8_t)(t + 1)), z(32, 2), m(32, 3), ek(ekl), dk(dkl), c(cl), K(32), K2(32);
                int32_t st = 1;
                if (mlkem_keygen(768, 1, d.data(), z.data(), ek.data(), dk.data())) bad++;
                for (int i = 0; i < 20; i++) { mlkem_encaps(768, 1, ek.data(), m.data(), c.data(), K.data()); mlkem_decaps(768, 1, dk.data(), c.data(), K2.data(), &st); }
                ready++;
                while (!go.load()) std::this_thread::yield();
                const double t0 = now_s();
                for (int i = 0; i < R; i++) {
                    m[0] = (uint8_t)i;
                    if (mlkem_encaps(768, 1, ek.data(), m.data(), c.data(), K.data())) bad++;
                    if (mlkem_decaps(768, 1, dk.data(), c.data(), K2.data(), &st)) bad++;
                    if (st != 0 || memcmp(K.data(), K2.data(), 32)) bad++;
                }
                secs[t] = now_s() - t0;
            });
        while (ready.load() < T) std::this_thread::yield();
        go = true;
        for (auto& x : th) x.join();
        double wall = 0;
        for (double s : secs) wall = s > wall ? s : wall;
        printf("MLKEM_HOST_LANES=%s MLKEM_HOST_COMBINE=%s threads=%2d: %8.0f pairs/s  (%.1f us per pair and thread, errors %d)\n", lanes ? lanes : "default", getenv("MLKEM_HOST_COMBINE") ? getenv("MLKEM_HOST_COMBINE") : "default", T, T * R / wall,
               wall / R * 1e6, bad.load());
    }
    mlkem_host_release();
    return 0;
}

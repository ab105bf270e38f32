// CPU harness for the ED pruner launcher's model of a launch (kv-match_amd/csrc/kvm_ed_launch_model.h; compiled by
// tests/test_abi.py with g++ -Wall -Werror and the sanitizers).  Prints one line per check; exit code = failed checks.
#include "kvm_ed_launch_model.h"

#include <chrono>
#include <cstdio>

static int failed = 0;
#define CHECK(cond, ...)                     \
    do {                                     \
        const bool ok_ = (cond);             \
        printf("%s ", ok_ ? "ok  " : "FAIL"); \
        printf(__VA_ARGS__);                 \
        printf("\n");                        \
        if (!ok_) failed++;                  \
    } while (0)

int main() {
    const double slots = 2048;
    int fa = 0, fb = 0;
    // the bench list of BASELINE configs[1] (m = 8192; 1 856 / 533 / 346 units): measured best (1, 4), 0.468 ms
    double t = edd_best_split(8192, 1856, 533, 346, slots, fa, fb);
    CHECK(fa == 1 && fb == 4 && t > 350 && t < 600, "bench list: split %d,%d modelled %.0f us", fa, fb, t);
    // short windows, many units: parts are mostly prologue, no split (measured: 1,1 0.548 ms against 1,4 0.767 ms)
    t = edd_best_split(1024, 4027, 7275, 10251, slots, fa, fb);
    CHECK(fa == 1 && fb == 1, "m = 1024, 21 553 units: split %d,%d modelled %.0f us", fa, fb, t);
    // a list that cannot fill the chip: split everything (measured: 4,4 best at 390 + 34 units)
    t = edd_best_split(8192, 390, 34, 0, slots, fa, fb);
    CHECK(fa == 4 && fb >= 2, "m = 8192, 424 units: split %d,%d modelled %.0f us", fa, fb, t);
    // more work never ends earlier under a fixed split
    bool mono = true;
    double prev = 0;
    for (int u = 0; u <= 40000; u += 250) {
        const double v = edd_model(8192, u, u / 3.0, u / 5.0, 1, 2, slots);
        if (v < prev - 1e-6) mono = false;
        prev = v;
    }
    CHECK(mono, "monotone in the number of units");
    // tile classes: no class-C tiles -> 1; more than one unit per wave slot -> 2; a small list of short runs -> 2
    CHECK(edd_choose_classes(8192, 1000, 500, 0, 256) == 1, "no remainders of <= 16: classes A / B");
    CHECK(edd_choose_classes(8192, 14842, 4262, 5522, 256) == 2, "bench list: classes A / B / C");
    CHECK(edd_choose_classes(8192, 1600, 3000, 19000, 256) == 2, "200 000 candidates in runs of ~8: classes A / B / C");
    CHECK(edd_choose_classes(8192, 1600, 3000, 19000, 256) == 2, "(memoised) the same again");
    // class A on 16 candidates per lane: by the size of the list (profiles/r05_a16_sweep.log)
    CHECK(edd_use_a16(8192, 931 + 533 + 344, slots), "bench list (0.88 of the wave slots in 16-tile units): 16 per lane");
    CHECK(!edd_use_a16(8192, 150, slots) && edd_use_a16(8192, 195 + 17 + 9, slots) && !edd_use_a16(2048, 195 + 17 + 9, slots) &&
              edd_use_a16(2048, 466 + 267 + 170, slots),
          "tiny lists stay with 8 per lane: below 0.1 of the wave slots in units, 0.25 for m < 4096");
    t = edd_best_split(8192, 931, 533, 344, slots, fa, fb, 1);
    CHECK((fa == 2 || fa == 1) && t > 300 && t < 600, "bench list, 16 per lane: split %d,%d modelled %.0f us", fa, fb, t);
    // ... and class B on the four-lane rows as well (form 2, 16-tile units; measured 0.385 - 0.40 ms alone)
    t = edd_best_split(8192, 928, 267, 347, slots, fa, fb, 2);
    CHECK(t > 300 && t < 600, "bench list, four-lane forms: split %d,%d modelled %.0f us", fa, fb, t);
    // several queries in flight: the launcher hands the model its share of the wave slots; the long units of a list of many rounds are not split
    (void)edd_best_split(8192, 928, 267, 347, slots / 8, fa, fb, 2);
    CHECK(fa == 1 && fb <= 2, "bench list on an eighth of the wave slots: split %d,%d", fa, fb);
    // degenerate inputs end, with finite results
    const double z = edd_model(8192, 0, 0, 0, 4, 4, slots);
    CHECK(z == 0.0, "no units: %.1f", z);
    const double h = edd_model(64, 1e6, 1e6, 1e6, 4, 4, slots);
    CHECK(std::isfinite(h) && h > 0, "three million units: %.0f us", h);
    const double ns = edd_model(8192, 100, 100, 100, 1, 1, 0.0);
    CHECK(ns == ns || true, "no wave slots: returns (%.1f)", ns);
    // cost of a decision the launcher cannot take from its memo: nine simulations
    const auto t0 = std::chrono::steady_clock::now();
    double acc = 0;
    const int reps = 2000;
    for (int r = 0; r < reps; r++) acc += edd_best_split(8192, 1856 + r % 7, 533, 346, slots, fa, fb);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    CHECK(us < 200.0 && acc > 0, "one decision (9 simulations): %.2f us", us);
    const auto t1 = std::chrono::steady_clock::now();
    for (int r = 0; r < 200; r++) acc += edd_best_split(512, 30000 + r, 1000, 1500, slots, fa, fb);
    const double us2 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / 200;
    CHECK(us2 < 2000.0, "one decision at 32 500 units: %.2f us", us2);
    return failed;
}

// kvm_ed_launch_model.h -- host-only: the ED pruner launcher's model of a launch (kvm_ed_dot.hip).  Plain C++ (no HIP), so
// that tests/harness/launch_model_harness.cpp can compile it with g++ (tests/test_abi.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>

#ifndef ED4_JC  // (kvm_internal.h defines them for the library; the harness includes this header alone)
#define ED4_JC 160
#define ED8_JC 128
#define ED16_JC 80
#endif
#define EDM_TILES_A 8   // tiles per unit (= per wavefront): classes A and B one per row, class C two
#define EDM_TILES_A16 16  // class A in its 16-candidates-per-lane form (every-point evaluation): 16 rows of 4 lanes
#define EDM_TILES_B16 16  // ... and class B with it: 8 candidates per lane on the same four-lane rows
#define EDM_TILES_B 8
#define EDM_TILES_C 16

// ---- the launcher's model of a launch: which window split (fA for class A, fB for classes B and C) ends first -----------
// Fitted to profiles/ed_split_sweep.py (81 list shapes x 9 split pairs, m = 512 .. 8192, 10^5 .. 4 10^6 candidates, mean
// run 8 .. 400), by profiles/fit_ed_launch_model.py.  Round 2's fit (profiles/r02_ed_split_sweep_{a,b}.log): rms error 7 %.
// Refitted late in round 3 (profiles/r03_split_sweep_{a,b}.log) -- every class now works in chunks of 128 points and a part's
// constant fell from 6-9 to 4-6 us (the whole-window sums of a lane's candidates are loaded together): rms 8.5 %, the pair it
// picks is 3.4 % (geometric mean; worst 20 %) behind the best measured pair, 4.3 % (worst 31 %) with round 2's constants.
//  * a part (one wave's share of a unit) costs  a + b ceil(chunks / f)  microseconds on a full chip -- the per-part
//    prologue (statistics, staging tables, first chunk) is what makes splitting short windows expensive;
//  * parts start in launch order (A, C, B) on `slots` wave slots;
//  * every resident wave advances at a speed that depends on how full the chip is (two waves on a SIMD share its fp64
//    pipe; below one wave per SIMD the clock is higher as well): piecewise linear in the fraction of occupied slots.
// The makespan is then a short event simulation over cohorts of identical parts (a few dozen events).
static inline double edd_speed(double x) {
    static const double kx[6] = {1.0, 0.75, 0.55, 0.5, 0.25, 0.0}, kv[6] = {1.0, 1.093, 1.149, 1.687, 1.695, 2.240};
    if (x >= 1.0) return 1.0;
    for (int i = 1; i < 6; i++)
        if (x >= kx[i]) return kv[i - 1] + (kv[i] - kv[i - 1]) * (kx[i - 1] - x) / (kx[i - 1] - kx[i]);
    return kv[5];
}
static inline double edd_makespan(const double cnt_in[3], const double w[3], double slots) {
    constexpr int CAP = 48;
    double cnt[3] = {cnt_in[0], cnt_in[1], cnt_in[2]};
    double a_cnt[CAP], a_rem[CAP];
    int na = 0, qi = 0;
    double free_slots = slots, t = 0.0;
    for (int guard = 0; guard < 100000; guard++) {
        while (free_slots > 0 && qi < 3) {
            if (cnt[qi] <= 0) { qi++; continue; }
            const double take = std::min(free_slots, cnt[qi]);
            if (na == CAP) {  // (never seen: cohorts retire as fast as they start) -- finish with the fluid bound
                double work = 0;
                for (int i = 0; i < na; i++) work += a_cnt[i] * a_rem[i];
                for (int c = qi; c < 3; c++) work += cnt[c] * w[c];
                return t + work / slots;
            }
            a_cnt[na] = take, a_rem[na] = w[qi], na++;
            free_slots -= take, cnt[qi] -= take;
        }
        if (na == 0) return t;
        double n = 0, rem = 1e300;
        for (int i = 0; i < na; i++) n += a_cnt[i], rem = std::min(rem, a_rem[i]);
        t += rem / edd_speed(n / slots);
        int k = 0;
        for (int i = 0; i < na; i++) {
            a_rem[i] -= rem;
            if (a_rem[i] <= 1e-9) free_slots += a_cnt[i];
            else a_cnt[k] = a_cnt[i], a_rem[k] = a_rem[i], k++;
        }
        na = k;
    }
    return t;
}
// modelled duration (microseconds, without the launch's constant) of unitsA / unitsB / unitsC units split fA / fB ways
// Class A in its 16-per-lane form (kvm_ed_dot.hip, EDD16_*: 16 rows x 4 lanes, the Karatsuba split of a pair's correlation):
// when?  profiles/ed_a16_sweep.py -> profiles/r05_a16k_sweep.log, 36 list shapes, both forms under their best split pair: the
// 16-per-lane form is 5 - 19 % ahead from 0.1 of the wave slots in units (16-tile class-A units + class-B + class-C units) on
// for windows of 4096 points and more, and 4 - 15 % ahead from 0.25 of the slots on for shorter windows; below that it is 3 - 6 %
// behind (half as many, longer parts, and a part's prologue fills a 128-point ring per row).
static inline int64_t edd_a16_min_units(int m, double slots) { return (int64_t)std::ceil((m >= 4096 ? 0.1 : 0.25) * slots); }
static inline bool edd_use_a16(int m, int64_t units16, double slots) { return units16 >= edd_a16_min_units(m, slots); }
// (form 0: 8 per lane.  form 1: the class-A units are 16-tile units of the 16-per-lane form; its two constants fitted by
// profiles/fit_ed_a16.py to the same sweep, the rest of the model as it was: rms log error 6 %.  form 2: class B as well on
// the four-lane rows, 8 per lane, in 16-tile units, launched before class C (its units are the longer ones); its two
// constants fitted to profiles/r05_b8k_sweep.log with the others held: rms log error 7 %, the pair picked 2 % behind the best)
static inline double edd_model(int m, double unitsA, double unitsB, double unitsC, int fA, int fB, double slots, int form = 0) {
    auto chunks = [](int mm, int jc, int f) { return std::ceil(std::ceil((double)mm / jc) / f); };
    const double wA = form ? 10.693 + 7.331 * chunks(m, ED8_JC, fA) : 5.188 + 4.535 * chunks(m, ED8_JC, fA),
                 wC = 6.360 + 3.899 * chunks(m, ED8_JC, fB),
                 wB = form == 2 ? 13.139 + 5.252 * chunks(m, ED8_JC, fB) : 3.951 + 3.719 * chunks(m, ED8_JC, fB);
    auto parts = [](double units, int f) { return 4.0 * std::ceil(units * f / 4.0); };  // whole workgroups of 4 waves
    if (form == 2) {   // launch order with the four-lane forms: A, B, C
        const double w2[3] = {wA, wB, wC}, cnt2[3] = {parts(unitsA, fA), parts(unitsB, fB), parts(unitsC, fB)};
        return edd_makespan(cnt2, w2, slots);
    }
    const double w[3] = {wA, wC, wB}, cnt[3] = {parts(unitsA, fA), parts(unitsC, fB), parts(unitsB, fB)};   // launch order: A, C, B
    return edd_makespan(cnt, w, slots);
}

// smallest modelled duration over the nine split pairs
static inline double edd_best_split(int m, double unitsA, double unitsB, double unitsC, double slots, int &fA, int &fB, int form = 0) {
    double best = 1e300;
    fA = fB = 1;
    for (int a = 1; a <= 4; a *= 2)
        for (int b = 1; b <= 4; b *= 2) {
            const double t = edd_model(m, unitsA, unitsB, unitsC, a, b, slots, form);
            if (t < best * (1.0 - 1e-3)) best = t, fA = a, fB = b;   // (near ties: the coarser split)
        }
    return best;
}

// Tiling of a host-planned list: 2 = classes A / B / C, 1 = remainders of <= 16 candidates in class-B tiles as well.  Class
// C fills the lanes better (a remainder of <= 16 takes 19 us of a wave at m = 8192 instead of 30) but its units last
// longer: on a list that does not fill the wave slots once the model decides (profiles/ed_hybrid_sweep.py: with the
// model's splits class C is 1.0-1.5 x faster on lists of short runs and within +-2 % elsewhere, 4 shapes of 90 lose 5-10 %).
static inline int edd_choose_classes(int m, int64_t tilesA, int64_t tilesB, int64_t tilesC, int num_cu, bool a16 = false) {   // (a16: class A in 16-tile units, model form 1)
    const double slots = (double)num_cu * 8;
    const int tpa = a16 ? EDM_TILES_A16 : EDM_TILES_A;
    const int64_t uA = (tilesA + tpa - 1) / tpa, uB = (tilesB + EDM_TILES_B - 1) / EDM_TILES_B,
                  uC = (tilesC + EDM_TILES_C - 1) / EDM_TILES_C, uBC = (tilesB + tilesC + EDM_TILES_B - 1) / EDM_TILES_B;
    if (tilesC <= 0) return 1;
    if ((double)(uA + uB + uC) > slots) return 2;  // throughput-bound: fewer, fuller tiles
    struct memo_t { int64_t a, b, c; int cu, m, mode; bool a16; };
    static thread_local memo_t memo = {-1, -1, -1, 0, 0, 1, false};
    if (memo.a == tilesA && memo.b == tilesB && memo.c == tilesC && memo.cu == num_cu && memo.m == m && memo.a16 == a16) return memo.mode;
    int fa, fb;
    const double t1 = edd_best_split(m, (double)uA, (double)uBC, 0.0, slots, fa, fb, a16 ? 1 : 0);
    const double t2 = edd_best_split(m, (double)uA, (double)uB, (double)uC, slots, fa, fb, a16 ? 1 : 0);
    const int mode = t2 <= t1 * 1.02 ? 2 : 1;
    memo = {tilesA, tilesB, tilesC, num_cu, m, mode, a16};
    return mode;
}


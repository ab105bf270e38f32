// CPU harness for the library's host-side index code (kv-match_amd/csrc/kvm_index_file.cpp): step 2, the IndexNode codec
// and the file layout run here WITHOUT a GPU, under AddressSanitizer / UBSan.  The GPU part -- step 1,
// kvm_index_rows_build_grouped -- is replaced by rows read from a file the test wrote (the oracle's step-1 rows, grouped);
// the image kvm_index_build produces from them goes to the output file and the test compares it with the oracle's.
//   usage: index_file_harness <rows.bin> <image.out> [team_min_rows team_min_row_size team_size]
//   rows.bin: int64 count, int64 groups, double keys[count], int32 first[count], int32 last[count],
//             double group_keys[groups], int32 group_offsets[groups + 1]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/kvmatch.h"

struct kvm_index_rows {
    std::vector<double> keys, gkeys;
    std::vector<int32_t> first, last, goff;
};
static kvm_index_rows g_rows;
static std::string g_err;

void kvm_set_error(kvm_ctx *, const std::string &msg) { g_err = msg; }
void kvm_set_error_noexcept(kvm_ctx *, const char *msg) noexcept { try { g_err = msg; } catch (...) {} }

int kvm_index_rows_build_for_file(kvm_ctx *, int32_t, kvm_index_rows **out) {   // (the library's internal step-1 call)
    *out = &g_rows;
    return KVM_OK;
}
extern "C" {
int64_t kvm_index_rows_groups(const kvm_index_rows *r) { return (int64_t)r->gkeys.size(); }
int64_t kvm_index_rows_count(const kvm_index_rows *r) { return (int64_t)r->keys.size(); }
const double *kvm_index_rows_group_keys(const kvm_index_rows *r) { return r->gkeys.data(); }
const int32_t *kvm_index_rows_group_offsets(const kvm_index_rows *r) { return r->goff.data(); }
const double *kvm_index_rows_keys(const kvm_index_rows *r) { return r->keys.data(); }
const int32_t *kvm_index_rows_first(const kvm_index_rows *r) { return r->first.data(); }
const int32_t *kvm_index_rows_last(const kvm_index_rows *r) { return r->last.data(); }
void kvm_index_rows_free(kvm_index_rows *) {}
}

template <class T>
static bool rd(FILE *f, std::vector<T> &v, size_t n) {
    v.resize(n);
    return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

void kvm_index_file_test_thresholds(int64_t rows, int64_t row_size, int64_t team_size);   // kvm_index_file.cpp: which ranges of rows a team of threads walks, and of how many

int main(int argc, char **argv) {
    if (argc != 3 && argc != 6) return 2;
    if (argc == 6) kvm_index_file_test_thresholds(atoll(argv[3]), atoll(argv[4]), atoll(argv[5]));   // (small lists through the pipelined walk as well)
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t cnt = 0, groups = 0;
    if (fread(&cnt, 8, 1, f) != 1 || fread(&groups, 8, 1, f) != 1) return 2;
    if (!rd(f, g_rows.keys, (size_t)cnt) || !rd(f, g_rows.first, (size_t)cnt) || !rd(f, g_rows.last, (size_t)cnt) ||
        !rd(f, g_rows.gkeys, (size_t)groups) || !rd(f, g_rows.goff, (size_t)groups + 1))
        return 2;
    fclose(f);
    kvm_index_file *img = nullptr;
    const int rc = kvm_index_build((kvm_ctx *)(uintptr_t)1, 0, &img);
    if (rc) {
        fprintf(stderr, "kvm_index_build: %d %s\n", rc, g_err.c_str());
        return rc == KVM_ERR_ILLEGAL_ARGUMENT ? 3 : 4;
    }
    printf("%lld %lld %lld %lld\n", (long long)kvm_index_file_info(img, KVM_INDEX_ROWS_BEFORE),
           (long long)kvm_index_file_info(img, KVM_INDEX_ROWS_AFTER), (long long)kvm_index_file_info(img, KVM_INDEX_INTERVALS_BEFORE),
           (long long)kvm_index_file_info(img, KVM_INDEX_INTERVALS_AFTER));
    const int wrc = kvm_index_file_write(img, argv[2]);
    kvm_index_file_free(img);
    return wrc ? 5 : 0;
}

// et_options.hip -- et_set_option / et_get_option (include/eigentraj.h): the one translation unit with mutable
// configuration.  See et_options.h for what each switch selects.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "et_common.h"
#include "et_options.h"

namespace et {
Options &options() {
    static Options o;
    return o;
}
}  // namespace et

namespace {
struct Key {
    const char *name;
    enum { kInt, kI64, kChar } kind;
    void *field;
    const char *allowed;  // kChar: the first letters accepted
};
const Key *keys(int *n) {
    et::Options &o = et::options();
    static const Key table[] = {
        {"kmeans_packed_min", Key::kI64, &o.kmeans_packed_min, nullptr},
        {"kmeans_argmax", Key::kChar, &o.kmeans_argmax, "fv"},
        {"kmeans_packed", Key::kInt, &o.kmeans_packed, nullptr},
        {"kmeans_init_tiles", Key::kInt, &o.kmeans_init_tiles, nullptr},
        {"kmeans_pack_fused", Key::kInt, &o.kmeans_pack_fused, nullptr},
        {"kmeans_filter_threads", Key::kInt, &o.kmeans_filter_threads, nullptr},
        {"kmeans_chain_copies", Key::kInt, &o.kmeans_chain_copies, nullptr},
        {"kmeans_loop_grid", Key::kInt, &o.kmeans_loop_grid, nullptr},
        {"kmeans_loop", Key::kChar, &o.kmeans_loop, "acp"},
        {"reforder_filter_min_lp", Key::kInt, &o.reforder_filter_min_lp, nullptr},
        {"reforder_init_skip_min", Key::kI64, &o.reforder_init_skip_min, nullptr},
        {"reforder_single_update", Key::kInt, &o.reforder_single_update, nullptr},
        {"metrics_form", Key::kChar, &o.metrics_form, "atf"},
    };
    *n = (int)(sizeof(table) / sizeof(table[0]));
    return table;
}
const Key *find(const char *name) {
    int n = 0;
    const Key *t = keys(&n);
    for (int i = 0; i < n; ++i)
        if (name && !strcmp(name, t[i].name)) return &t[i];
    return nullptr;
}
}  // namespace

extern "C" int et_set_option(const char *key, const char *value) {
    const Key *k = find(key);
    if (!k || !value || !value[0]) return ET_ERR_INVALID_ARG;
    if (k->kind == Key::kChar) {
        if (!strchr(k->allowed, value[0])) return ET_ERR_INVALID_ARG;
        static_cast<std::atomic<int> *>(k->field)->store(value[0], std::memory_order_relaxed);
        return ET_OK;
    }
    char *end = nullptr;
    const long long v = strtoll(value, &end, 0);
    if (end == value || *end) return ET_ERR_INVALID_ARG;
    if (k->kind == Key::kI64) static_cast<std::atomic<int64_t> *>(k->field)->store((int64_t)v, std::memory_order_relaxed);
    else static_cast<std::atomic<int> *>(k->field)->store((int)v, std::memory_order_relaxed);
    return ET_OK;
}

extern "C" int et_get_option(const char *key, char *value, size_t value_bytes) {
    const Key *k = find(key);
    if (!k || !value || value_bytes < 2) return ET_ERR_INVALID_ARG;
    if (k->kind == Key::kChar) {
        value[0] = (char)static_cast<std::atomic<int> *>(k->field)->load(std::memory_order_relaxed);
        value[1] = 0;
    } else if (k->kind == Key::kI64) {
        snprintf(value, value_bytes, "%lld", (long long)static_cast<std::atomic<int64_t> *>(k->field)->load(std::memory_order_relaxed));
    } else {
        snprintf(value, value_bytes, "%d", static_cast<std::atomic<int> *>(k->field)->load(std::memory_order_relaxed));
    }
    return ET_OK;
}

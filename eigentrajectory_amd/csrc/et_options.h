// et_options.h -- the library's process-wide tuning switches (include/eigentraj.h: et_set_option / et_get_option).
//
// Every switch is a MEASUREMENT AID or a test lever: the defaults are the shipped configuration, results are bit for bit
// the same for every setting (each selects between forms that are tested equal), and nothing here is read from the
// environment -- this translation unit pair is the only place of the library that holds mutable configuration.
#pragma once

#include <atomic>
#include <stdint.h>

namespace et {

struct Options {
    std::atomic<int64_t> kmeans_packed_min{0};   // >= 1024: shards of at least this many points iterate on the packed copy (default 131072 = 2^17)
    std::atomic<int> kmeans_argmax{'f'};          // 'f': matrix-core filter + exact certification; 'v': the exact scan only
    std::atomic<int> kmeans_packed{1};            // 0: trace-less fits keep the fp32 filter
    std::atomic<int> kmeans_init_tiles{1};        // 0: farthest-first steps look at every point's running similarity
    std::atomic<int> kmeans_pack_fused{1};        // 0: the packed copy is written by a pass of its own before the loop
    std::atomic<int> kmeans_filter_threads{0};    // 256 .. 1024 (multiple of 64): threads per workgroup of the Lloyd kernels; 0: chosen per shard
    std::atomic<int> kmeans_chain_copies{2};      // 1, 2, 4, 8: compact copies of the delta table the chained Lloyd kernel of a single-GPU fit adds its deltas onto
    std::atomic<int> kmeans_loop_grid{0};         // > 0: at most this many workgroups for the chained Lloyd kernel (grid sweep); 0: one resident round
    std::atomic<int> kmeans_loop{'a'};            // 'a'uto, 'c'hain (one launch per iteration), 'p'ersist (one launch per fit)
    std::atomic<int> reforder_filter_min_lp{9};  // reference-order Lloyd: level power from which the matrix-core label filter is used (4: always, 9: never = default: it measured slower)
    std::atomic<int64_t> reforder_init_skip_min{(int64_t)1 << 21};  // reference-order farthest-first: shards of at least this many points test (bestR, nearest) before reading a point's coordinates
    std::atomic<int> reforder_single_update{1};  // reference-order Lloyd on shards of few level-2 blocks: 1: one 1024-thread workgroup does levels 2, 3 and the update; 0: the grid of block workgroups + last arriver
    std::atomic<int> metrics_form{'a'};           // 'a'uto, 't'ile (vector-ALU workgroup-tile kernel), 'f' (fp32 matrix instructions)
};

Options &options();

}  // namespace et

// cf_knobs.hpp — the library's environment knobs, behind ONE gate.
//
// The CF_* variables below are a debug / experiment interface: the tests force table combinations and kernel variants with them
// (tests/test_async_abi.py, tests/test_gpu_scale.py), tools/gpu/run.sh measures alternatives.  They are read ONLY while
// CF_DEBUG_KNOBS=1 is set as well (tests/conftest.py and the tools set it): a stray CF_* variable in a user's environment changes
// nothing.  The supported ways to say the same things are cf_index_options (tables), cf_params and the command line.
//
//   index tables (override cf_index_options): CF_WIDE_FTAB, CF_TEXT_VERIFY_RATE, CF_ISA_RATE, CF_OCC_PLANES, CF_DENSE_SA_RATE, CF_PAIR_PLANES,
//       CF_DROP_SIDES, CF_MULTI_VERIFY (small_range_rows), CF_MULTI_MIN_RUN, CF_TEXT_VERIFY_MIN_RUN, CF_TABLE_PLANNER (0 = the
//       fixed priorities of rounds 2 - 3), CF_FORCE_WIDE_SIDE (64-bit side division on a small index), CF_RESTORE_SHIFT, CF_RESTORE_VERBOSE
//   kernel variants: CF_SEARCH_V, CF_WALK_V, CF_BLOCKS_PER_CU, CF_LAZY_N, CF_LAZY_HITS, CF_SELF_RECORDS, CF_REV_WORDS, CF_POST_FAST, CF_SCORE_FAST,
//       CF_DIRECT_REFS, CF_POS_HITS, CF_POST_LDS, CF_SCORE_LDS_ROWS, CF_SCORE_LDS_SPARSE, CF_TAIL_STREAM, CF_EARLY_SCORE, CF_COUNT_SLOT_BITS, CF_ROWS_PER_QUERY
//   builder: CF_BUILD_ROUNDS, CF_BUILD_DOUBLING          front end: CF_TEST_FAIL_OPEN, CF_TEST_FAIL_COMM (error paths of the multi-GPU driver), CF_CLI_RCCL, CF_CLI_PACKED, CF_DUMP_FROM_PACKED, CF_INGEST_BLOCK, CF_INGEST_STREAM,
//       CF_CLI_DEVICE_TEXT (0 = the parser pool for every input), CF_CLI_TEXT_HOST_PARSE (1 = every text block through the host parser), CF_TEXT_BLOCK (bytes per text block), CF_CLI_MAP_OUTPUT
#pragma once
#include <cstdlib>

namespace cfamd {
inline const char *cf_knob(const char *name) {
    static const bool on = [] { const char *v = std::getenv("CF_DEBUG_KNOBS"); return v && *v && std::atoi(v) != 0; }();
    return on ? std::getenv(name) : nullptr;
}
}  // namespace cfamd

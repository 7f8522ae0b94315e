/* subset_main.c — runs the reference's own unit-test functions (compiled in place from /root/reference/tests/*.c, see the
 * Makefile) against libzxc_mi355x.so. Test infrastructure. The case list is the public-API part of the reference's table
 * (tests/test_main.c:60-64 Block API, :50-58 Buffer API, :66-69 contexts, :144-172 seekable, :90-104 push streaming, :72-77 static contexts, :120-141 dictionaries, the FILE* stream cases); each function returns 1 on
 * success like there. Usage: zxc_unit_subset [--list] [name-substring]. Prints "RESULT name PASS|FAIL" per case. */
#include <stdio.h>
#include <string.h>
#include "test_common.h"

typedef int (*test_fn_t)(void);
typedef struct { const char* name; test_fn_t fn; } test_entry_t;
static const test_entry_t g_tests[] = {
    TEST_CASE(test_buffer_api), TEST_CASE(test_buffer_api_scratch_buf), TEST_CASE(test_buffer_error_codes),
    TEST_CASE(test_decompress_inplace), TEST_CASE(test_get_decompressed_size), TEST_CASE(test_decompress_fast_vs_safe_path),
    TEST_CASE(test_max_compressed_size_logic), TEST_CASE(test_decompress_empty_frame_null_dst),
    TEST_CASE(test_block_api), TEST_CASE(test_block_api_boundary_sizes), TEST_CASE(test_block_api_large_block_varint),
    TEST_CASE(test_decompress_block_bound), TEST_CASE(test_decompress_block_safe),
    TEST_CASE(test_opaque_context_api), TEST_CASE(test_cctx_level_raise_reinit), TEST_CASE(test_estimate_cctx_size),
    TEST_CASE(test_error_name), TEST_CASE(test_library_info_api),
    TEST_CASE(test_seekable_table_sizes), TEST_CASE(test_seekable_table_write), TEST_CASE(test_seekable_roundtrip),
    TEST_CASE(test_seekable_open_query), TEST_CASE(test_seekable_random_access), TEST_CASE(test_seekable_non_seekable_reject),
    TEST_CASE(test_seekable_single_block), TEST_CASE(test_seekable_all_levels), TEST_CASE(test_seekable_many_blocks),
    TEST_CASE(test_seekable_open_file), TEST_CASE(test_seekable_open_reader),
    TEST_CASE(test_seekable_mt_roundtrip), TEST_CASE(test_seekable_mt_single_block), TEST_CASE(test_seekable_mt_random_access),
    TEST_CASE(test_seekable_mt_full_file), TEST_CASE(test_seekable_open_reader_mt),
    TEST_CASE(test_seekable_cross_boundary), TEST_CASE(test_seekable_truncated_input), TEST_CASE(test_seekable_corrupted_sek),
    TEST_CASE(test_seekable_range_out_of_bounds), TEST_CASE(test_seekable_dst_too_small), TEST_CASE(test_seekable_empty_file),
    TEST_CASE(test_seekable_no_checksum), TEST_CASE(test_seekable_with_checksum), TEST_CASE(test_seekable_work_buf_tail_pad),
    /* push streaming, tests/test_main.c:90-104 */
    TEST_CASE(test_pstream_roundtrip_basic), TEST_CASE(test_pstream_roundtrip_no_checksum), TEST_CASE(test_pstream_roundtrip_levels),
    TEST_CASE(test_pstream_tiny_chunks), TEST_CASE(test_pstream_drip_one_byte), TEST_CASE(test_pstream_empty_input),
    TEST_CASE(test_pstream_large_random), TEST_CASE(test_pstream_compatible_with_buffer_api),
    TEST_CASE(test_pstream_decompress_compatible_with_buffer_api), TEST_CASE(test_pstream_invalid_args),
    TEST_CASE(test_pstream_truncated_input), TEST_CASE(test_pstream_corrupted_magic), TEST_CASE(test_pstream_decode_seekable_archive),
    TEST_CASE(test_pstream_compress_after_end_rejected), TEST_CASE(test_pstream_compress_drain_block_resume),
    /* static contexts, tests/test_main.c:72-77 */
    TEST_CASE(test_static_ctx_size_query), TEST_CASE(test_static_ctx_workspace_too_small), TEST_CASE(test_static_ctx_block_size_locked),
    TEST_CASE(test_static_ctx_level_raise_rejected), TEST_CASE(test_static_ctx_null_inputs), TEST_CASE(test_static_ctx_roundtrip_all_levels),
    /* dictionaries incl. training, tests/test_main.c:120-141 (its .zxd cases build a table with two internal helpers: huf_helpers.c) */
    TEST_CASE(test_dict_zxd_roundtrip), TEST_CASE(test_dict_id_deterministic), TEST_CASE(test_dict_get_id_apis), TEST_CASE(test_dict_buffer_roundtrip),
    TEST_CASE(test_dict_block_roundtrip), TEST_CASE(test_dict_block_safe_roundtrip), TEST_CASE(test_dict_mismatch_error), TEST_CASE(test_dict_required_error),
    TEST_CASE(test_dict_no_dict_compat), TEST_CASE(test_dict_stream_roundtrip), TEST_CASE(test_dict_large_dict_roundtrip), TEST_CASE(test_dict_safe_loop_backref),
    TEST_CASE(test_dict_seekable_roundtrip), TEST_CASE(test_dict_train_roundtrip), TEST_CASE(test_dict_train_no_frequent_patterns),
    TEST_CASE(test_dict_seekable_mt_roundtrip), TEST_CASE(test_dict_stream_dict_id_checks), TEST_CASE(test_dict_seekable_dict_id_checks),
    TEST_CASE(test_dict_huf_zxd_roundtrip), TEST_CASE(test_dict_huf_table_roundtrip), TEST_CASE(test_dict_huf_degenerate_corpus),
    /* the FILE* stream API, tests/test_main.c (test_stream_api.c) */
    TEST_CASE(test_null_output_decompression), TEST_CASE(test_invalid_arguments), TEST_CASE(test_truncated_input), TEST_CASE(test_io_failures),
    TEST_CASE(test_thread_params), TEST_CASE(test_multithread_roundtrip), TEST_CASE(test_stream_get_decompressed_size_errors),
    TEST_CASE(test_stream_engine_errors), TEST_CASE(test_roundtrip_offset_mixed),
};

int main(int argc, char** argv) {
    zxc_test_srand(42); /* the reference's fixed seed (tests/test_main.c:200) */
    const char* filter = NULL;
    const size_t n = sizeof(g_tests) / sizeof(g_tests[0]);
    if (argc > 1 && strcmp(argv[1], "--list") == 0) {
        for (size_t i = 0; i < n; i++) printf("%s\n", g_tests[i].name);
        return 0;
    }
    if (argc > 1) filter = argv[1];
    int failed = 0, ran = 0;
    for (size_t i = 0; i < n; i++) {
        if (filter && !strstr(g_tests[i].name, filter)) continue;
        const int ok = g_tests[i].fn();
        printf("RESULT %s %s\n", g_tests[i].name, ok ? "PASS" : "FAIL");
        fflush(stdout);
        ran++;
        failed += !ok;
    }
    printf("SUMMARY ran %d failed %d\n", ran, failed);
    return failed != 0;
}

#!/bin/bash
# Runs the GPU parity suite under non-default tunables (each must give the same answers).
# Tests that assert on path-specific statistics are deselected.
out=$GRAFT_REPO_ROOT/gpurun_out/option_sweep.log
: > $out
for opts in "serialize_scans=0" "queries_per_launch=1" "queries_per_launch=3,blocks_per_cu=1" "shape_kernels=0,block_threads=128" \
            "mq_fused=0,mq_i8=0" "mq_tail_overlap=1,mq_blocks=2" "multi_query=0,query_batch=5" "contexts=1,blocks_per_cu=6" "mask_dense=0,coalesce=0" "mq_min=8,tie_mode=0"; do
  echo "== $opts" >> $out
  SZG_OPTIONS=$opts timeout -k 10 600 python -m pytest tests -m gpu -q -x \
      --deselect tests/test_gpu_multiquery.py::test_shared_sweep_matches_oracle \
      --deselect tests/test_gpu_multiquery.py::test_shared_sweep_quantized_rows \
      --deselect tests/test_gpu_multiquery.py::test_shared_sweep_euclidean \
      --deselect tests/test_gpu_multiquery.py::test_shared_sweep_euclidean_far_from_origin \
      --deselect tests/test_gpu_multiquery.py::test_shared_sweep_int8_mfma \
      --deselect tests/test_gpu_multiquery.py::test_fused_selection_overflow_falls_back \
      --deselect tests/test_gpu_collection.py::test_concurrent_single_queries_are_coalesced \
      -k "not two_shards and not masks_tombstones" 2>&1 | tail -3 >> $out
done
cat $out
# shared-sweep variants on the shared-sweep tests (their statistics do not depend on these)
for opts in "mq_fused=0" "mq_i8=0" "mq_tail_overlap=1" "mq_fused=0,mq_tail_overlap=1,serialize_scans=0"; do
  echo "== multiquery tests, $opts" >> $out
  SZG_OPTIONS=$opts timeout -k 10 600 python -m pytest tests/test_gpu_multiquery.py -q -x \
      --deselect tests/test_gpu_multiquery.py::test_fused_selection_overflow_falls_back 2>&1 | tail -2 >> $out
done
tail -12 $out

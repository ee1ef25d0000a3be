#!/bin/bash
# Run the GPU test families in separate processes (a faulting kernel must not hide the other results).
# Usage (on the GPU box): bash tools/gpu_suite.sh [extra pytest args]
mkdir -p gpurun_out
export TMPDIR=/tmp
LOG=gpurun_out/gpu_suite.log
: > $LOG
run() {
  echo "=== $* ===" >> $LOG
  timeout 600 python -m pytest "$@" -m gpu -q --tb=short -p no:cacheprovider >> $LOG 2>&1
  echo "exit=$?" >> $LOG
}
for t in test_conv2d_fwd_bwd test_convT2d_fwd_bwd test_linear_fwd_bwd test_linear_small_batch_paths test_input_expansion_bit_exact test_head_softmax test_poe_reparam_kl test_poe_draws_its_own_noise \
         test_bce_and_ce test_lincomb_rows test_embed_pe test_attention test_layernorm_residual \
         test_time_reduce_and_permute_mask test_adam_amsgrad_flat_matches_torch test_txt_layer_fused_matches_op_by_op test_txt_layer_with_pooled_heads test_lprob_rowsum test_optimal_sigma_rowsum test_conv_generic test_reduce_segments test_seeded_losses; do
  run tests/test_hip_ops.py -k $t
done
run tests/test_parity_e2e.py -k golden
run tests/test_parity_e2e.py -k full_size -s
run tests/test_parity_e2e.py -k graph_replay
run tests/test_parity_e2e.py -k dropout
run tests/test_parity_e2e.py -k three_modalities
run tests/test_parity_e2e.py -k captured_step_every_mixer
run tests/test_parity_e2e.py -k checkpoint_round_trip
run tests/test_parity_e2e.py -k load_batch_compact
run tests/test_parity_e2e.py -k forward_with_missing
grep -E "^===|passed|failed|error|exit=" $LOG | tail -60

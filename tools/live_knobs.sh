# sourced by the A/B and sweep scripts: the MMVAE_* environment variables the library / package still read.
# A knob that is not in this list is no longer read anywhere: comparing two of its values compares two identical
# configurations, so the scripts refuse it.  Keep in sync with DESIGN.md "Appendix: environment knobs"
# (tests/test_abi.py::test_live_knob_list_matches_the_sources checks it against the sources).
LIVE_KNOBS="MMVAE_POE_BALANCE MMVAE_EARLY_ADAM MMVAE_ATTN_T_BWD MMVAE_FFN32_LN MMVAE_PROJ32_LN MMVAE_SKINNY_DW MMVAE_RING_SPLIT MMVAE_LINEAR_DW_LATER_AT MMVAE_GEMM_B16 MMVAE_LINEAR_DW_LATER MMVAE_TXT_WAVE MMVAE_TXT_WAVE_BWD_MIN_N MMVAE_TXT_WAVE_BWD_MIN_N_DEC MMVAE_STREAMS MMVAE_DP_STAGED MMVAE_DP_OVERLAP MMVAE_GRAPH_COLLECTIVE MMVAE_INPUT_PIPE_DMA MMVAE_MARKS MMVAE_GRAPH_DUMP MMVAE_HIP_LIB MMVAE_RESNET50_WEIGHTS"
require_live_knob() {
  for k in $LIVE_KNOBS; do [ "$k" = "$1" ] && return 0; done
  echo "knob $1 is not read by the library or the package any more (live: $LIVE_KNOBS)" >&2
  exit 2
}

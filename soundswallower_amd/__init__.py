"""MI355X-native SoundSwallower acoustic hot path: GMM senone scoring + forced alignment.

The compute lives in `libssw_amd.so` (C host code + hand-written gfx950 HIP kernels) behind the
C ABI of `include/ssw_amd.h`; this package is the thin host mirror used by tests and bench.
"""
from .api import (CompactPlan, FirstPassPlan, INT_MAX, SCORER_MS, SCORER_PTM, Lexicon, Model, MsMgau, PtmMgau,
                  StateAlignSearch, SswError, Texts, align_text_batch, align_text_batch_active, forced_align_batch, forced_align_planned, forced_alignment, model_dir)
from .synth import lcg_uniform, synth_features, synth_alignment_task

__all__ = ["Model", "CompactPlan", "PtmMgau", "MsMgau", "Lexicon", "StateAlignSearch", "SswError", "FirstPassPlan", "Texts", "align_text_batch", "align_text_batch_active", "forced_align_batch", "forced_align_planned", "forced_alignment", "model_dir", "SCORER_PTM",
           "SCORER_MS", "INT_MAX", "lcg_uniform", "synth_features", "synth_alignment_task"]

"""ORACLE -- CPU restatement of the reference's hot path. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package, and only
as the checker. The product path (whisper-char-alignment_amd/) never imports it and has no CPU fallback.

Parity status:
  * reference-owned code (timing.py filter_attention / aggregation / jump arithmetic, retokenize.py,
    metrics.py) -- PINNED against the real reference files executed under stub modules
    (tests/golden/make_golden.py -> tests/golden/*.npz).
  * upstream openai-whisper arithmetic (log-mel, model forward, median_filter, dtw_cpu/backtrace, and the greedy
    decoding loop with its logit filters in decoding_ref.py) -- PARITY UNPINNED: the package is an unpinned,
    un-vendored dependency that is absent offline; restated from its published algorithm and cross-checked against
    HuggingFace transformers' independent implementations (model forward, feature extractor, and its ports of
    dtw / median_filter / the timestamp logit rules: tests/test_oracle.py).
"""

"""CPU oracle for the GCA contrastive pre-training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: it may be
imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the timed CPU baseline,
never as the thing shipped.  The product path (``video-graph-ssl_amd``) never
imports it and fails loudly when the HIP library is missing.

What it is: a plain PyTorch-CPU fp32 restatement of the reference's algorithm
for the path ``tools/train_video_contrast_dis.py::_train_moco/_train_simsiam``
(encoder -> graph block -> head -> MoCo queue / InfoNCE), each function citing
the reference file:line it follows.  The arithmetic itself lives in ATen
(third party, no version pinned by the reference; this image: torch 2.10 CPU).

Parity pin: the reference has no tests / golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned by fixtures generated HERE by
importing the reference's own classes (``tests/golden/make_golden.py``, run in
the build container where ``/root/reference`` exists) and committed as small
``.npz`` files under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks
every oracle module against them.
"""

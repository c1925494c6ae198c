"""CPU oracle for the MoRec in-batch train step -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch CPU restatement (numpy for the integer bookkeeping,
PyTorch-CPU fp32 for the floating-point maths) of the algorithm on the hot path of
``inbatch_sasrec_e2e_text`` in westlake-repl/IDvs.MoRec.  Every function cites the reference
file:line it follows.  It exists to CHECK the HIP path; nothing in the shipped package
(``idvs.morec_amd``) imports it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.

Pinning: the oracle is checked against golden vectors captured from the *imported reference*
(``tests/golden/make_golden.py`` run once in the build container, fixtures committed under
``tests/golden/``) by ``tests/test_oracle_vs_golden.py``.  The BERT arithmetic lives in the
third-party ``transformers`` package (reference pins 4.20.1, ``README.md:44``; the container has
5.15.0, eager attention) -- the goldens therefore pin the installed-HF behaviour.
"""
from .bookkeeping import (  # noqa: F401
    collate_train_sample,
    ce_labels,
    column_valid,
    reject_mask,
    valid_rows,
    log_pop,
    pooled_targets,
)
from .nn_ref import (  # noqa: F401
    sasrec_forward,
    bert_forward,
    text_encoder_forward,
    inbatch_ce_loss,
    model_forward,
)
from .optim_ref import adamw_step  # noqa: F401
from .data_ref import read_behaviors_ref, read_news_ref  # noqa: F401
from .eval_ref import eval_ranks, hit_ndcg_at_k  # noqa: F401
from .swin_ref import SwinCfg, swin_forward, vit_encoder_forward  # noqa: F401
from .bce_ref import bce_loss, bce_model_forward  # noqa: F401

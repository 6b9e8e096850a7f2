"""Generation options of the hot path — field-for-field
``SequenceGeneratorOptions`` of the reference
(src/seamless_communication/inference/generator.py:59-84)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List,  Any, Optional, Tuple


class NGramRepeatBlockProcessor:
    """The step processor the reference's CLI installs for ``--text_generation_ngram_blocking``
    (cli/m4t/predict/predict.py:172-175; class from fairseq2 0.2 ``fairseq2.generation`` — not under
    /root/reference, restated from its published behaviour: parity unpinned).

    Called with ``seqs`` (rows, S) = the sequences generated so far (prompt included) and ``probs``
    (rows, V): every token that would complete an n-gram already present in its row is blocked
    (``-inf`` for log-probabilities, ``0`` for probabilities).  On the HIP path only ``ngram_size`` is
    read (``sc_gen_opts.no_repeat_ngram_size``); ``__call__`` is the host restatement for other callers.
    """

    def __init__(self, ngram_size: int) -> None:
        if ngram_size <= 0:
            raise ValueError("`ngram_size` must be greater than 0.")
        self.ngram_size = int(ngram_size)

    def __call__(self, seqs, probs, lprob: bool = False) -> None:
        import torch

        g = self.ngram_size
        rows, seq_len = seqs.shape
        if g >= seq_len:
            return
        fill = -torch.inf if lprob else 0.0
        if g == 1:
            probs.scatter_(1, seqs.to(torch.int64), fill)
            return
        windows = seqs.unfold(1, g, 1)  # (rows, S-g+1, g)
        tail = seqs[:, seq_len - g + 1:]  # (rows, g-1)
        for r in range(rows):
            hit = (windows[r, :, :-1] == tail[r]).all(dim=1)
            probs[r, windows[r, hit, -1].to(torch.int64)] = fill


@dataclass
class SequenceGeneratorOptions:
    """Holds the options to pass to a sequence generator."""

    beam_size: int = 5
    """The beam size."""

    soft_max_seq_len: Tuple[int, int] = (1, 200)
    """The terms ``a`` and ``b`` of ``ax + b`` where ``x`` is the source
    sequence length. The generated sequences (including prefix sequence) will
    have the maximum length of ``min(hard_max_seq_len, ax + b)``."""

    hard_max_seq_len: int = 1024
    """The hard limit on maximum length of generated sequences."""

    step_processor: Optional[Any] = None
    """The processor called at each generation step."""

    unk_penalty: float = 0.0
    """The UNK symbol penalty."""

    len_penalty: float = 1.0
    """The length penalty (beam search only)."""


def remove_consecutive_repeated_ngrams(sequence: List[int], min_size: int = 1, max_size: int = 40) -> List[int]:
    """Unit post-filter of the autoregressive T2U, TRANSLITERATED from the reference's function of the same name
    (inference/generator.py:39-56, used at :355-362 when ``unit_generation_ngram_filtering`` is set, batch size 1 only;
    host integer logic, same control flow): scanning from the left, an n-gram (longest first) that is immediately followed
    by a copy of itself loses its first copy."""
    assert 1 <= min_size <= max_size
    drop = set()
    start = 0
    while start < len(sequence):
        for k in range(max_size, min_size - 1, -1):
            if sequence[start: start + k] == sequence[start + k: start + 2 * k]:
                drop.update(range(start, start + k))
                start += k - 1
                break
        start += 1
    return [tok for i, tok in enumerate(sequence) if i not in drop]

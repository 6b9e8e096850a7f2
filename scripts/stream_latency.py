#!/usr/bin/env python3
"""Per-segment latency of the streaming S2ST chain (BASELINE cfg 5) on one MI355X: full-size synthetic weights
(seamlessM4T_v2_large shapes + the dense_1b monotonic decoder), one 10 s utterance fed in 320 ms segments with the
settings of the reference's cli/streaming/evaluate.py:56-69.  Prints one JSON object per policy variant.

Synthetic weights never emit EOS and their p_choose values are not those of a trained model, so the read/write pattern
(and with it the per-segment decoder work) is illustrative: `decision_method=min` over 24x16 heads reads until the
source ends, `mean` alternates reads and writes."""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="base_v2", choices=["base_v2", "tiny_v2"])
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--max-len-b", type=int, default=100, help="cli/streaming/evaluate.py:64; lower it for tiny_v2 (unit_max_seq_len)")
    a = ap.parse_args()
    from seamless_communication_amd import cards, synthetic as syn
    from seamless_communication_amd.config import seamless_m4t_v2_large, tiny_config
    from seamless_communication_amd.runtime import HipS2STModel
    from seamless_communication_amd.streaming import HipStreamingBackend, SeamlessStreamingS2STAgent, SpeechSegment, default_args
    from seamless_communication_amd.tokenizer import CharTokenizer, NllbTextTokenizer

    cfg = seamless_m4t_v2_large() if a.arch == "base_v2" else tiny_config()
    t0 = time.perf_counter()
    tt = NllbTextTokenizer(cfg.text_vocab_size, cards.TEXT_LANGS)
    ct = CharTokenizer(cfg.char_vocab_size)
    model = HipS2STModel(cfg, syn.make_unity_state_dict(cfg), syn.make_vocoder_state_dict(cfg), device=0,
                         monotonic_state_dict=syn.make_monotonic_decoder_state_dict(cfg))
    model.set_nar_tables(tt, ct)
    load_s = time.perf_counter() - t0
    be = HipStreamingBackend(model, cfg)
    stage = {}
    for name in ("fbank", "encode_speech", "mma_begin", "mma_step", "t2u", "vocode"):
        orig = getattr(be, name)

        def timed(*args, _o=orig, _n=name, **kw):
            t = time.perf_counter()
            r = _o(*args, **kw)
            torch.cuda.synchronize()
            stage[_n] = stage.get(_n, 0.0) + (time.perf_counter() - t)
            stage[_n + "_calls"] = stage.get(_n + "_calls", 0) + 1
            return r

        setattr(be, name, timed)
    wav = syn.synthetic_waveform(0, a.seconds).numpy()
    seg = 5120
    for method in ("min", "mean"):
        args = default_args(tgt_lang="fra", min_starting_wait_w2vbert=192, decision_threshold=0.5, no_early_stop=True, max_len_a=0,
                            max_len_b=a.max_len_b, min_unit_chunk_size=50, decision_method=method)
        for rep in range(2):  # first pass warms allocations up
            agent = SeamlessStreamingS2STAgent(be, tt, args)
            stage.clear()
            ms, out_samples, pos = [], 0, 0
            while pos < len(wav):
                chunk = wav[pos: pos + seg]
                pos += seg
                s = SpeechSegment(content=chunk.tolist(), sample_rate=16000, finished=pos >= len(wav), tgt_lang="fra")
                t = time.perf_counter()
                out = agent.pushpop(s)
                torch.cuda.synchronize()
                ms.append((time.perf_counter() - t) * 1e3)
                if not out.is_empty:
                    out_samples += len(out.content)
                if out.finished:
                    break
        tokens = len(agent.module_list[2].states.target_indices)
        print(json.dumps({
            "metric": "streaming S2ST wall time per 320 ms source segment", "arch": a.arch, "decision_method": method,
            "segments": len(ms), "p50_ms": float(np.percentile(ms, 50)), "p90_ms": float(np.percentile(ms, 90)),
            "max_ms": float(max(ms)), "total_s": float(sum(ms) / 1e3), "audio_s": a.seconds, "rtf": float(sum(ms) / 1e3 / a.seconds),
            "text_tokens_written": tokens, "output_audio_s": out_samples / 16000.0,
            "stage_s": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in sorted(stage.items())},
            "load_seconds": round(load_s, 1), "data": "synthetic weights and audio",
        }))


if __name__ == "__main__":
    main()

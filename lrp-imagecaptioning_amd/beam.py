"""Beam bookkeeping of the reference's caption search, separated from where the scores come from.

Reference: `ExplainImgCaptioningAttentionModel._beam_search` (models/explainers.py:51-120) with
`BatchNLargest` / `NLargest` / `Caption` (inference.py:267-315).  What it does, per image:

* the live set ("partial captions") is the `beam_size` LARGEST candidates under the total order
  (log_prob, sentence) — a bounded heap of `Caption(log_prob, sentence_encoded)` tuples.  EVERY candidate goes
  in, the ones that just produced EOS included (E:88-93): such a hypothesis keeps its beam slot and is extended
  past its EOS in the following steps like any other;
* a candidate whose new word is EOS is ALSO recorded as a complete caption — the sentence *before* the word was
  appended — with the candidate's log-probability (E:94-98), in a second bounded heap;
* per step every live hypothesis contributes its `beam_size` most probable next words
  (`np.argpartition(preds, -beam_size)`, E:76-78) under a log-soft-max (E:45-48);
* after `max_caption_length` steps the answer is the best complete caption if there is one, else the best live
  one, without the leading SOS and with the trailing EOS (E:108-119).

`search` reproduces exactly that, for several images at once, on top of a `step` callable that yields the
top-k continuations of every live row — the device path (lrp_decoder_gen_step + lrp_op_log_softmax_topk), the
replay path and the canned scores of the CPU parity test all plug in there.  Pinned against the reference's own
`_beam_search` run on canned scores: tests/golden/beam_*.npz, tests/test_beam.py.
"""
import numpy as np


def search(step, n_images, beam_size, max_caption_length, eos):
    """step(s, parent, word) -> (ids, logp): arrays (n_images * beam_size, beam_size); row i*k + j belongs to
    beam j of image i.  For s > 0 row r continues the hypothesis that lived in row parent[r] at the previous step
    with tokenizer id word[r] appended; ids are MODEL columns (tokenizer id - 1, E:92), logp their log-probabilities.
    Returns, per image, up to `beam_size` captions (word ids + [eos]): complete ones by descending score, then live ones —
    element [0] is what the reference returns for that image."""
    k = beam_size
    beams = [[((), 0.0)] for _ in range(n_images)]        # (words, log_prob), descending — `n_largest(sort=True)`
    rows = [[0] for _ in range(n_images)]                  # row (within the image's k) each live beam was computed in
    complete = [[] for _ in range(n_images)]
    for s in range(max_caption_length):
        if s == 0:
            ids, logp = step(0, None, None)
        else:
            parent, word = [], []
            for i in range(n_images):
                pad = k - len(beams[i])                    # rows without a hypothesis replay beam 0 (results ignored)
                parent += [i * k + r for r in rows[i]] + [i * k + rows[i][0]] * pad
                word += [b[0][-1] for b in beams[i]] + [beams[i][0][0][-1]] * pad
            ids, logp = step(s, parent, word)
        ids, logp = np.asarray(ids), np.asarray(logp, dtype=np.float64)
        for i in range(n_images):
            cand = []
            for r, (words, lp) in enumerate(beams[i]):
                for c, l in zip(ids[i * k + r], logp[i * k + r]):
                    w = int(c) + 1                         # model column -> tokenizer id (E:92)
                    total = lp + float(l)
                    cand.append((words + (w,), total, r))
                    if w == eos:
                        complete[i].append((words, total))
            # the k largest under (log_prob, sentence) — what a bounded heap of Caption tuples retains (inference.py:297-309)
            cand.sort(key=lambda c: (c[1], c[0]), reverse=True)
            complete[i].sort(key=lambda c: (c[1], c[0]), reverse=True)
            del complete[i][k:]
            keep = cand[:k]
            beams[i] = [(c[0], c[1]) for c in keep]
            rows[i] = [c[2] for c in keep]
    out = []
    for i in range(n_images):
        caps = [list(c[0]) + [eos] for c in complete[i][:k]]
        caps += [list(b[0]) + [eos] for b in beams[i][:k - len(caps)]]
        out.append(caps)
    return out


def topk_log_softmax(logits, k):
    """Host form of the ranking step (E:45-48, E:76-78) for (rows, V) scores: (ids, logp) by descending probability —
    used by the replay path and the CPU tests; the device path is lrp_op_log_softmax_topk."""
    x = np.asarray(logits)
    x = x - x.max(axis=-1, keepdims=True)
    lp = x - np.log(np.exp(x).sum(axis=-1, keepdims=True))
    ids = np.argsort(-lp, axis=-1, kind="stable")[:, :k]
    return ids, np.take_along_axis(lp, ids, axis=-1)

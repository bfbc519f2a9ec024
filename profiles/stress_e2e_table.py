#!/usr/bin/env python3
"""Format the records tests/test_gpu_stress_parity.py::test_trained_like_weights_through_the_decoder appends to
gpurun_out/parity.jsonl into the seeds x modes table of profiles/r04_stress_e2e.txt.
Usage: python profiles/stress_e2e_table.py [parity.jsonl] > profiles/r04_stress_e2e.txt"""
import json
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity.jsonl"
ref, rows = {}, []
for ln in open(path):
    d = json.loads(ln)
    if d.get("name") == "stress_e2e_reference":
        ref[d["seed"]] = d
    elif str(d.get("name", "")).startswith("stress_decoder_") and "seed" in d:
        rows.append(d)
print("# tests/test_gpu_stress_parity.py::test_trained_like_weights_through_the_decoder on an MI355X box (relative L1 of the raw")
print("# (224,224,3) heat-map, worst over the explained tokens).  Trained-like VGG16 kernels (5 % dense, lognormal sigma 1.5,")
print("# >= 80 % dead activations) -> adaptive decoder -> CNN LRP; four draws of (kernels, decoder weights, caption).")
print("# 'float32 reference' = float32 C.forward -> AdaptiveOracle (numpy) -> float32 literal graph on the box's CPU: what TF/numpy compute.")
print("# 'undecidable' = units (location, channel) of relu(F.W_if + b_if) (E:378-381) whose float64 pre-activation is below 1e-5 of")
print("# sum|F W| + |b|; 'flipped' = units the evaluator decided differently from the float64 pipeline; 'aligned' = distance to the")
print("# float64 pipeline with exactly those decisions taken over (the assertion: < 1e-4).")
print()
for s in sorted(ref):
    r = ref[s]
    print("seed %d  tokens %s  undecidable units %s  smallest relative margin %.2e" % (s, r["tokens"], r["undecidable_units"], r["min_margin"]))
    print("   %-10s %-12s %-12s %-12s %-12s %s" % ("evaluator", "features", "end to end", "aligned", "given feat.", "flipped units"))
    print("   %-10s %-12s %-12.2e %-12s %-12s %s" % ("float32 ref", "-", r["float32_pipeline"], "-", "-", r["float32_pipeline_flipped_units"]))
    for d in rows:
        if d["seed"] == s:
            print("   %-10s %-12.2e %-12.2e %-12.2e %-12.2e %s" % (d["name"].replace("stress_decoder_", ""), d["features"],
                  d["end_to_end_vs_float64_pipeline"], d["end_to_end_decisions_aligned"], d["given_engine_features"], d["flipped_units"]))
    print()

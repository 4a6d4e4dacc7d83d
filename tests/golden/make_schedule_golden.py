#!/usr/bin/env python3
"""Golden vectors for trainer.temperature_proportion / temperature: imports the reference's
steps/libs/nnet3/train/temperature_schedule.py (it depends on nothing but `logging`) IN THIS CONTAINER and records what
its two functions return.  Usage: python tests/golden/make_schedule_golden.py"""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_temperature_schedule", "/root/reference/steps/libs/nnet3/train/temperature_schedule.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

G = {"proportion": [], "adapt": []}
for k in range(0, 41):
    f = k / 40.0
    G["proportion"].append({"data_fraction": f, "edit": ref.get_temperature_edit_string(f, 7)})
for t0, t1 in ((1.0, 0.03), (5.0, 0.5), (0.8, 0.8)):
    for k in range(0, 11):
        f = k / 10.0
        G["adapt"].append({"init": t0, "final": t1, "data_fraction": f, "edit": ref.get_temperature_edit_string_adapt(t0, t1, f, 3)})
G["adapt_none"] = ref.get_temperature_edit_string_adapt(None, 0.5, 0.3, 0)
json.dump(G, open(os.path.join(HERE, "r01_schedule_golden.json"), "w"), indent=0)
print(G["proportion"][3], G["adapt"][4])

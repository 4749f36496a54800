"""Writes tests/golden/reference_config_keys.json: the flattened key -> default-value map of the reference's
src/config.yaml (data read from /root/reference, run in the build container).  tests/test_host_cpu.py checks that this
build's src/config.yaml carries every one of those keys with the same type and default."""
import json
import os

import yaml


def flatten(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(flatten(v, prefix + k + "."))
        else:
            out[prefix + k] = v
    return out


if __name__ == "__main__":
    with open("/root/reference/src/config.yaml") as f:
        ref = flatten(yaml.safe_load(f))
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "reference_config_keys.json"), "w") as f:
        json.dump(ref, f, indent=1, sort_keys=True)
    print(len(ref), "keys")

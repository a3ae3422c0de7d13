#!/usr/bin/env python3
"""Capture the reference's argparse Namespace (after the YAML overlay) for the three cfg presets and
for a CLI case exercising the `type=bool` quirk -> tests/golden/opts_namespaces.json.
Build container only (imports /root/reference)."""
import json
import os
import sys
import types

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/anet-video-captioning"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
tb = types.ModuleType("tensorboardX")
tb.SummaryWriter = object
sys.modules["tensorboardX"] = tb
import opts as ref_opts  # noqa: E402
from misc.utils import update_values  # noqa: E402

out = {}
for name in ("baseline", "cyclical", "code_development"):
    sys.argv = ["main.py", "--path_opt", "cfgs/%s.yml" % name]
    o = ref_opts.parse_opt()
    with open(os.path.join(REF, "cfgs", name + ".yml")) as h:
        update_values(yaml.safe_load(h), vars(o))
    out[name] = vars(o)
sys.argv = ["main.py", "--train_decoder_only", "False", "--resume", "0", "--beam_size", "3", "--cuda"]
out["cli_quirk"] = vars(ref_opts.parse_opt())
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "opts_namespaces.json"), "w"), indent=1, sort_keys=True)
print({k: len(v) for k, v in out.items()})

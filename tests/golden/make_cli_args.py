"""The CTC parameter set of BASELINE config 2 (ctc-common + ctc-all-intra + longdress_vox10 + ctc-r3) as a flat list of
PccAppEncoder command-line options: the reference's cfg files are `key: value` lines, and every key is also an option of its
command line.  Lets the reference's CLI run where the reference tree (and with it cfg/) is absent -- bench.py's cpu_baseline
leg on the GPU box.  Data only (parameter values); paths and frame counts are left to the caller.
usage: python tests/golden/make_cli_args.py  ->  tests/golden/cli_ctc_args.json"""
import json
import os

REF = "/root/reference/cfg"
FILES = ["common/ctc-common.cfg", "condition/ctc-all-intra.cfg", "sequence/longdress_vox10.cfg", "rate/ctc-r3.cfg"]
CALLER = {"uncompressedDataPath", "frameCount", "startFrameNumber", "groupOfFramesSize",     # the caller's business
          # paths of HDRTools configurations: only checked for existence (the build has no HDRTools: internal converter)
          "colorSpaceConversionConfig", "inverseColorSpaceConversionConfig"}
args = {}
for f in FILES:
    for line in open(os.path.join(REF, f)):
        line = line.split("#")[0].strip()
        if not line or ":" not in line:
            continue
        k, v = line.split(":", 1)
        if k.strip() not in CALLER:
            args[k.strip()] = v.strip()          # later files override earlier ones, as on the command line
out = {"source": FILES, "args": ["--%s=%s" % kv for kv in args.items()]}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cli_ctc_args.json"), "w"), indent=1)
print(len(out["args"]), "options")

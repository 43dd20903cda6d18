"""Fixture of the reference CLI: every `parser.add_argument(...)` of /root/reference/train_mobody.py as data
(flag, default, type name, action), extracted from the source text with `ast` (nothing is executed).
Run here (the reference is not present on the GPU box): python tests/golden/make_cli_golden.py"""
import ast
import json
import os

REF = "/root/reference/train_mobody.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g10_cli_flags.json")


def lit(node):
    try:
        return ast.literal_eval(node)
    except Exception:
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == "int":
            return int(ast.literal_eval(node.args[0]))                      # default=int(1e6)
        return ast.unparse(node)


flags = []
for node in ast.walk(ast.parse(open(REF).read())):
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
        kw = {k.arg: k.value for k in node.keywords}
        flags.append(dict(flag=lit(node.args[0]), default=lit(kw["default"]) if "default" in kw else None,
                          type=ast.unparse(kw["type"]) if "type" in kw else None,
                          action=lit(kw["action"]) if "action" in kw else None))
json.dump(sorted(flags, key=lambda f: f["flag"]), open(OUT, "w"), indent=1)
print(len(flags), "flags ->", OUT)


# ---- the merged config dict (train_mobody.py:470-531): every key of the `config.update({...})` literal and the
# expression it is bound to, as data; plus the yaml files of the MOBODY configs the reference ships (parsed values)
import glob

import yaml

OUT2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g10b_config_merge.json")
tree = ast.parse(open(REF).read())
merge = None
for node in ast.walk(tree):
    if (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "update"
            and isinstance(node.func.value, ast.Name) and node.func.value.id == "config" and node.args
            and isinstance(node.args[0], ast.Dict) and len(node.args[0].keys) > 20):
        merge = [(ast.literal_eval(k), ast.unparse(v)) for k, v in zip(node.args[0].keys, node.args[0].values)]
assert merge is not None
yamls = {}
for f in sorted(glob.glob("/root/reference/config/*/mobody/*.yaml")):
    yamls["/".join(f.split("/")[-3:])] = yaml.safe_load(open(f, encoding="utf-8"))
json.dump(dict(update=merge, yaml=yamls), open(OUT2, "w"), indent=1)
print(len(merge), "config keys,", len(yamls), "yaml files ->", OUT2)

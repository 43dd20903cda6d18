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

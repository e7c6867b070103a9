"""Build-container script: lists the names the reference's experiment scripts import from its ``src`` package
(``from src.x.y import a, b``), by parsing /root/reference/experiments/*.py with ``ast`` - NAMES ONLY, no source.
Output: tests/golden/experiment_imports.json, the fixture of tests/test_dropin_imports.py (SURVEY.md section 8(b),
"experiments drop in unchanged").  ``src.dmtet.*`` (DMTet geometry / rendering) is out of scope and listed apart."""
import ast
import glob
import json
import os

REF = "/root/reference/experiments"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    in_scope, out_of_scope = {}, {}
    for path in sorted(glob.glob(os.path.join(REF, "*.py"))):
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "src":
                dst = out_of_scope if node.module.startswith("src.dmtet") else in_scope
                for a in node.names:
                    dst.setdefault(node.module, {}).setdefault(a.name, []).append(os.path.basename(path))
    doc = {"generated_by": "tests/golden/make_import_list.py", "scripts": sorted(os.path.basename(p) for p in
                                                                                glob.glob(os.path.join(REF, "*.py"))),
           "in_scope": in_scope, "out_of_scope": out_of_scope}
    with open(os.path.join(HERE, "experiment_imports.json"), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print(json.dumps(doc, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()

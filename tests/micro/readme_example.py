"""Runs the code block of README.md's "Use" section (needs an MI355X)."""
import os, re
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "README.md")).read()
fence = "`" * 3
code = re.search(r"## Use\n\n" + fence + r"python\n(.*?)" + fence, src, re.S).group(1)
import sys; sys.path.insert(0, root)
ns = {}
exec(code, ns)
print("README example OK", float(ns["loss"]), tuple(ns["w"].shape), sorted(ns["traj"].keys()))

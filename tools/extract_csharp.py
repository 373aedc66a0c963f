#!/usr/bin/env python3
"""Write the C# shim of the integration document as files under bindings/csharp/ (one per section that holds csharp code),
so that a maintainer can copy them into PathTracer/ instead of cutting them out of the text.  The document is the source: run
this after editing it.  The files have not been compiled (no .NET toolchain here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = "// P/Invoke shim for the reference's C# host over libpt_hip.so (pt_hip.h): {title}.\n// Written by tools/extract_csharp.py; not compiled here (no .NET toolchain).\n"


def main():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    out = os.path.join(ROOT, "bindings", "csharp")
    os.makedirs(out, exist_ok=True)
    for title, body in re.findall(r"^## ([^\n]*)\n(.*?)(?=^## |\Z)", text, re.M | re.S):
        codes = re.findall(r"```csharp\n(.*?)```", body, re.S)
        if not codes:
            continue
        name = title.split()[0]                       # "Native.cs", "Wavefront.cs", "Scene.cs"
        stem, ext = os.path.splitext(name)
        path = os.path.join(out, stem + ("" if stem == "Native" else ".patch") + ext)
        with open(path, "w") as f:
            f.write(HEADER.format(title=title))
            f.write("\n".join(c.rstrip("\n") + "\n" for c in codes))
        print(path)


if __name__ == "__main__":
    main()

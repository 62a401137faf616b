"""Generates tests/golden/cdf_schema.json: the header of the reference's output and restart files -- global attributes,
dimensions, variables in definition order with their dimensions, type and EVERY attribute text -- read out of the
reference's own source (pom/io_pnetcdf.F: def_var_pnetcdf :6-40, write_output_pnetcdf :57-410, write_restart_pnetcdf
:1661-2083) by interpreting the statements that define it.  PnetCDF is absent from this image, so the reference cannot
write a file here; what it WOULD define is nevertheless fully determined by those statements.  The fixture is data (names,
texts, orders), not source text.  Run in a container that has /root/reference:

    python tests/golden/make_cdf_schema.py

Interpreted statements (fixed-form Fortran, continuation lines joined): `str_tmp='..'[//time_start]`, `length=<int|name>`,
`status=nfmpi_put_att_text(ncid,<nf_global|x_varid>,'<att>',length,<'text'|trim(str_tmp)|trim(title)>)`,
`status=nfmpi_def_dim(ncid,'<dim>',length,<x>_dimid)`, `vdims(k)=<x>_dimid`,
`call def_var_pnetcdf(ncid,'<name>',n,vdims,<x>_varid,<long_name>,<units>,<coords>,<.true.|.false.>)` with def_var_pnetcdf's
own body (nf_double; long_name, units, and coordinates when the flag is set).  Dimensions are recorded in the FILE's
(C / CDL) order, i.e. vdims reversed: Fortran's first index varies fastest.
"""
import json
import os
import re
import sys

REF = os.environ.get("POM_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "pom", "io_pnetcdf.F")


def statements(lines):
    """fixed form: join continuation lines (a non-blank, non-zero character in column 6), drop comments and cpp lines"""
    out = []
    for ln in lines:
        ln = ln.rstrip("\n")
        if not ln.strip() or ln[0] in "!cC*#":
            continue
        if len(ln) > 5 and ln[5] not in " 0" and ln[:5].strip() == "":
            out[-1] += ln[6:].strip()
        else:
            out.append(ln.strip())
    return out


def split_args(s):
    """arguments of one call, respecting quotes and parentheses"""
    args, cur, depth, q = [], "", 0, False
    for ch in s:
        if ch == "'":
            q = not q
        if not q and ch == "(":
            depth += 1
        if not q and ch == ")":
            depth -= 1
        if not q and depth == 0 and ch == ",":
            args.append(cur.strip())
            cur = ""
        else:
            cur += ch
    args.append(cur.strip())
    return args


def text_of(expr, env):
    """a character expression: 'literal', trim(x), x//y, str_tmp, title, time_start"""
    parts = []
    for piece in re.split(r"//", expr):
        piece = piece.strip()
        m = re.fullmatch(r"trim\((.*)\)", piece)
        if m:
            piece = m.group(1).strip()
        if piece.startswith("'") and piece.endswith("'"):
            parts.append(piece[1:-1].replace("''", "'"))
        elif piece == "str_tmp":
            parts.append(env["str_tmp"])
        elif piece in ("title", "time_start"):
            parts.append("{" + piece + "}")
        else:
            raise ValueError(f"cannot evaluate character expression {expr!r}")
    return "".join(parts)


def interpret(stmts):
    env = {"str_tmp": "", "length": None}
    gatts, dims, variables, vdims, varid = [], [], [], {}, {}
    for st in stmts:
        low = st.lower()
        m = re.fullmatch(r"str_tmp\s*=\s*(.*)", st)
        if m:
            env["str_tmp"] = text_of(m.group(1), env).rstrip()      # trim() of a blank-padded character*120
            continue
        m = re.fullmatch(r"length\s*=\s*(.*)", st)
        if m:
            v = m.group(1).strip()
            env["length"] = int(v) if v.isdigit() else v
            continue
        m = re.fullmatch(r"vdims\((\d)\)\s*=\s*(\w+)_dimid", st)
        if m:
            vdims[int(m.group(1))] = m.group(2)
            continue
        if low.startswith("status=nfmpi_def_dim("):
            a = split_args(st[st.index("(") + 1:st.rindex(")")])
            dims.append([a[1].strip("'"), env["length"]])
            continue
        if low.startswith("status=nfmpi_put_att_text("):
            a = split_args(st[st.index("(") + 1:st.rindex(")")])
            name, text = a[2].strip("'"), text_of(a[4], env)
            declared = env["length"] if isinstance(env["length"], int) else len(text)
            if a[1] == "nf_global":
                gatts.append([name, text])
            else:
                variables[varid[a[1]]]["atts"].append([name, text, declared])
            continue
        if low.startswith("call def_var_pnetcdf("):
            a = split_args(st[st.index("(") + 1:st.rindex(")")])
            name, n = a[1].strip("'"), int(a[2])
            v = {"name": name, "type": "double", "dims": [vdims[k] for k in range(n, 0, -1)],
                 "atts": [["long_name", text_of(a[5], env).rstrip(), None], ["units", text_of(a[6], env).rstrip(), None]]}
            if a[8].lower() == ".true.":
                v["atts"].append(["coordinates", text_of(a[7], env).rstrip(), None])
            for att in v["atts"]:
                att[2] = None if "{" in att[1] else len(att[1])     # def_var_pnetcdf: length = len(trim(text)); {..}: a run-time string
            varid[a[4]] = len(variables)
            variables.append(v)
            continue
        if low.startswith("status=nfmpi_enddef("):
            break
    return {"global_atts": gatts, "dims": dims, "vars": variables}


def routine(all_lines, name):
    start = next(n for n, ln in enumerate(all_lines) if re.match(rf"\s+subroutine {name}\b", ln))
    end = next(n for n in range(start + 1, len(all_lines)) if re.match(r"\s+end\s*$", all_lines[n]))
    return all_lines[start:end]


def main():
    if not os.path.exists(SRC):
        sys.exit(f"{SRC} not found: run where the reference is present")
    with open(SRC) as f:
        lines = f.readlines()
    out = {"source": "pom/io_pnetcdf.F: def_var_pnetcdf, write_output_pnetcdf, write_restart_pnetcdf (statements interpreted by tests/golden/make_cdf_schema.py)",
           "version_byte": 2,                                        # nf_clobber+nf_64bit_offset: CDF-2
           "output": interpret(statements(routine(lines, "write_output_pnetcdf"))),
           "restart": interpret(statements(routine(lines, "write_restart_pnetcdf")))}
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "cdf_schema.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k in ("output", "restart"):
        print(k, len(out[k]["vars"]), "variables,", len(out[k]["dims"]), "dimensions,", sum(len(v["atts"]) for v in out[k]["vars"]), "attributes")


if __name__ == "__main__":
    main()

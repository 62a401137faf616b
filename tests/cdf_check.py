"""TEST HELPER: the whole header of a file written by pomgpu_write_output / pomgpu_write_restart against
tests/golden/cdf_schema.json -- what the REFERENCE's own source defines (io_pnetcdf.F:57-410, :1661-2083, extracted by
tests/golden/make_cdf_schema.py): format version, global attributes, dimensions, and every variable in definition order with
its type, dimensions and every attribute text, in order."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def check_header(path, kind, title, time_start, kb, im_global, jm_global):
    from scipy.io import netcdf_file
    schema = json.load(open(os.path.join(HERE, "golden", "cdf_schema.json")))
    want = schema[kind]
    fill = lambda t: t.replace("{title}", title).replace("{time_start}", time_start)
    sizes = {"kb": kb, "kbm1": kb - 1, "im_global": im_global, "jm_global": jm_global}
    notes = []
    with netcdf_file(str(path), "r", mmap=False) as f:
        assert f.version_byte == schema["version_byte"], f.version_byte
        got_g = [[k, v.decode() if isinstance(v, bytes) else v] for k, v in f._attributes.items()]
        assert got_g == [[k, fill(v)] for k, v in want["global_atts"]], got_g
        got_d = [[k, v] for k, v in f.dimensions.items()]
        assert got_d == [[k, v if isinstance(v, int) else sizes[v]] for k, v in want["dims"]], got_d
        assert list(f.variables) == [v["name"] for v in want["vars"]], list(f.variables)
        for v in want["vars"]:
            g = f.variables[v["name"]]
            assert list(g.dimensions) == v["dims"], (v["name"], g.dimensions)
            assert g.typecode() == "d" and g.data.dtype.str == ">f8", (v["name"], g.typecode())
            got_a = [[k, a.decode() if isinstance(a, bytes) else a] for k, a in g._attributes.items()]
            assert got_a == [[k, fill(t)] for k, t, _ in v["atts"]], (v["name"], got_a)
            for k, t, declared in v["atts"]:
                if declared is not None and declared != len(t):
                    notes.append((v["name"], k, declared, len(t)))
    # the one place where the reference's header is not determined by its source: `formula_terms` of vtot is declared 26 bytes
    # long for the 10-byte literal 'time: time' (io_pnetcdf.F:140-142: it reads past the literal); the library writes the literal
    assert notes in ([], [("vtot", "formula_terms", 26, 10)]), notes
    return notes

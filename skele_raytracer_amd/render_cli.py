"""`raytracer` for one node of MI355Xs: the reference's command line (README.md:24-34, main.cpp:246-391), one process
per GPU.

    python -m skele_raytracer_amd.render_cli --path S.scn --output O.ppm [--width i] [--height i] [--fov f]
           [--gillum n] [--jsample g] [--depth d] [--parallel true|false] [--shadow] [--seed N] [--tile-rows r]
           [--strict-scn] [--shade-triangles] [--legacy-reflect] [--progressive K [--progressive-every M]] [--format ppm|png|pfm]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
           -m skele_raytracer_amd.render_cli --path spheres2.scn --output out.ppm --width 3840 --height 2160 \\
           --gillum 64 --jsample 5 --shadow            # BASELINE config 5

The framebuffer is cut into interleaved row tiles (tile t -> rank t mod G), every rank renders its tiles with one
`skr_render_tiles` launch sequence, ONE RCCL all-gather brings the u8 tiles to every rank and rank 0 de-interleaves
and writes the PPM (distributed.py).  The image does not depend on G: random numbers are keyed by the global pixel.
Flags are matched like the reference does — by `strcmp` anywhere in argv, unknown tokens ignored — and usage errors
print the reference's messages and exit with status 0 (main.cpp:381-391).  There is no CPU path.
"""
import os
import sys
import time


def _atoi(s):
    """C atoi: leading white space, an optional sign, digits; 0 when there are none (what main.cpp:246-379 calls)."""
    import re
    m = re.match(r"[ \t\n\v\f\r]*([+-]?\d+)", s)
    return max(-2 ** 31, min(2 ** 31 - 1, int(m.group(1)))) if m else 0


def _atof(s):
    """C atof: the longest leading decimal / hex-float / inf / nan prefix; 0.0 when there is none."""
    import re
    m = re.match(r"[ \t\n\v\f\r]*([+-]?(?:0[xX](?:[0-9a-fA-F]+\.?[0-9a-fA-F]*|\.[0-9a-fA-F]+)(?:[pP][+-]?\d+)?|(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?|inf(?:inity)?|nan))", s, re.I)
    if not m:
        return 0.0
    t = m.group(1)
    return float.fromhex(t) if "x" in t.lower() else float(t)


def _parse(argv):
    """main.cpp:246-379 as bin/raytracer restates it: flags anywhere, the value is the next token read with atoi / atof
    (`--width abc` is width 0, not a usage error); only a MISSING value is a usage error — and for --gillum a warning."""
    opt = dict(path=None, output=None, width=1920, height=1080, fov=60.0, gillum=None, jsample=None, depth=3, shadow=False, seed=1, tile_rows=8)
    given = set()  # --strict-scn: the .scn's film_resolution / max_depth hold for what the command line leaves open

    def value(i, kind, what):
        if i + 1 >= len(argv):
            raise ValueError(what)
        return kind(argv[i + 1])

    for i, a in enumerate(argv):
        if a == "--path":
            opt["path"] = value(i, str, "path must be passed after --path")
        elif a == "--output":
            opt["output"] = value(i, str, "output path must be passed after --output")
        elif a == "--width":
            opt["width"] = value(i, _atoi, "width takes an int after flag for the width")
            given.add("width")
        elif a == "--height":
            opt["height"] = value(i, _atoi, "height takes an int after flag for the width")
            given.add("height")
        elif a == "--fov":
            opt["fov"] = value(i, _atof, "fov takes a float (degrees) after flag for the field of view")
        elif a == "--gillum":
            if i + 1 < len(argv):  # main.cpp:250-253: monte_carlo = true, num_path_traces = atoi(next) — 0 for a non-number
                opt["gillum"] = _atoi(argv[i + 1])
            else:  # main.cpp:258 warns and goes on
                print("gillum takes an int after flag for the number of paths traced", file=sys.stderr)
        elif a == "--jsample":
            opt["jsample"] = value(i, _atoi, "jsample takes an int after flag for the supersampling grid size")
        elif a == "--depth":
            opt["depth"] = value(i, _atoi, "depth takes a positive int after flag for the max depth")
            if opt["depth"] <= 0:
                raise ValueError("depth takes a positive int after flag for the max depth")
            given.add("depth")
        elif a == "--shadow":
            opt["shadow"] = True
        elif a == "--strict-scn":
            opt["strict_scn"] = True
        elif a == "--shade-triangles":
            opt["shade_triangles"] = True
        elif a == "--legacy-reflect":
            opt["legacy_reflect"] = True
        elif a == "--progressive":
            opt["progressive"] = max(1, value(i, _atoi, "progressive takes the number of passes"))
        elif a == "--progressive-every":
            opt["progressive_every"] = max(0, value(i, _atoi, "progressive-every takes a number of passes"))
        elif a == "--format":
            opt["format"] = value(i, str, "format takes ppm, png or pfm")
        elif a == "--seed":
            opt["seed"] = value(i, _atoi, "seed takes an int")
        elif a == "--tile-rows":
            opt["tile_rows"] = value(i, _atoi, "tile-rows takes a positive int")
    if opt["tile_rows"] <= 0:
        raise ValueError("tile-rows takes a positive int")
    if opt.get("format", "ppm") not in ("ppm", "png", "pfm"):
        raise ValueError("format takes ppm, png or pfm")
    if opt.get("strict_scn"):
        opt["_given"] = given
    if opt["path"] is None:
        raise ValueError("no scene file was passed. Pass with --path path_to_scn")
    if opt["output"] is None:
        raise ValueError("no output destination was passed. Pass with --output destination_path.ppm")
    return opt


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    try:
        o = _parse(argv)
    except ValueError as e:
        print(str(e), file=sys.stderr)
        return 0  # the reference's usage errors leave with status 0
    if o["width"] <= 0 or o["height"] <= 0 or o["width"] > 65536 or o["height"] > 65536:
        print("raytracer: bad image size %dx%d" % (o["width"], o["height"]), file=sys.stderr)  # as bin/raytracer: before anything is sized from it
        return 2
    import torch
    import torch.distributed as dist
    import skele_raytracer_amd as skr
    from .distributed import FrameSharder

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise skr.SkrError("no MI355X visible: the renderer has no CPU path")
    # SKR_REHEARSE_GLOO=1: every rank on GPU 0, collectives over gloo (rehearsal of the N > 1 path on a one-GPU box)
    rehearsal = world > 1 and os.environ.get("SKR_REHEARSE_GLOO") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
    try:
        scene = skr.parse_scene(o["path"], strict=bool(o.get("strict_scn")))
    except skr.SkrError as e:
        if rank == 0:
            print(str(e))  # scene.cpp:24: "Can't open file" on stdout, exit(0)
        if world > 1:
            dist.destroy_process_group()
        return 0
    if o.get("strict_scn"):  # as bin/raytracer --strict-scn: the .scn's film_resolution and max_depth for what argv leaves open
        info = scene.info
        if "width" not in o["_given"] and info.film_width > 0:
            o["width"] = info.film_width
        if "height" not in o["_given"] and info.film_height > 0:
            o["height"] = info.film_height
        if "depth" not in o["_given"] and info.max_depth_parsed > 0:
            o["depth"] = info.max_depth_parsed
    r = skr.Renderer(scene, local_rank)
    kw = dict(fov=o["fov"], depth=o["depth"], shadow=o["shadow"], seed=o["seed"], shade_triangles=bool(o.get("shade_triangles")), progressive=o.get("progressive", 1), legacy_reflect=bool(o.get("legacy_reflect")))
    if o["gillum"] is not None:
        kw["gillum"] = o["gillum"]
    if o["jsample"] is not None:
        kw["jsample"] = o["jsample"]
    opt = skr.Options(o["width"], o["height"], **kw)
    fmt, every = o.get("format", "ppm"), o.get("progressive_every", 0)
    if world > 1 and (fmt == "pfm" or every):
        # the ranks exchange quantised tiles (one all-gather of bytes): the float frame and the running mean stay on their devices
        if rank == 0:
            print("raytracer: --format pfm and --progressive-every need the single-device path", file=sys.stderr)
        dist.destroy_process_group()
        return 2
    if fmt == "pfm" or every:
        # the file is the window (main.cpp:183-197 redraws its SDL window as rows finish): rewritten with the mean so far
        def write(rgb, rgbf):
            if fmt == "pfm":
                skr.write_pfm(o["output"], rgbf)
            else:
                (skr.write_png if fmt == "png" else skr.write_ppm)(o["output"], rgb)

        def progress(done, total, rgb, rgbf):
            if done < total:
                write(rgb, rgbf)
            print("pass %d of %d" % (done, total))
            return False

        opt = skr.Options(o["width"], o["height"], **kw)
        rgb, rgbf, ms = r.render_progressive_host(opt, every, want_float=(fmt == "pfm"), progress=progress if every else None)
        write(rgb, rgbf)
        print("***\nWROTE TO PPM\n***")  # main.cpp:213
        print("1 GPU(s), %dx%d, %.3f ms (device), kernel %s" % (o["width"], o["height"], ms, r.kernel_variant()), file=sys.stderr)
        return 0
    sharder = FrameSharder(o["width"], o["height"], o["tile_rows"], rank, world, dev)
    stream = torch.cuda.current_stream(dev)
    t0 = time.perf_counter()
    frame = sharder.step(lambda buf: r.render_tiles_into(opt, o["tile_rows"], rank, world, buf.data_ptr(), None, stream.cuda_stream))
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if rank == 0:
        (skr.write_png if fmt == "png" else skr.write_ppm)(o["output"], frame.cpu().numpy())
        print("***\nWROTE TO PPM\n***")  # main.cpp:213
        print("%d GPU(s), %dx%d, %.3f ms (render + gather), kernel %s" % (world, o["width"], o["height"], dt * 1e3, r.kernel_variant()), file=sys.stderr)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""AddressSanitizer + UBSan builds of the CPU-side code (the GPU pool does not offer sanitizers):
the host layer driven through its loaders with good and malformed input, and the oracle rendering
a small image."""
import os
import subprocess

import numpy as np

from conftest import ROOT

SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def test_host_layer_under_asan_ubsan(tmp_path):
    host = os.path.join(ROOT, "qaray_amd", "csrc", "host")
    srcs = [os.path.join(host, f) for f in ("xml.cpp", "scene.cpp", "mesh.cpp", "image.cpp", "xmlload.cpp", "framebuffer.cpp")]
    exe = str(tmp_path / "host_sanity")
    cmd = ["g++", "-std=c++17", "-pthread", *SAN, "-ffp-contract=off", f"-I{ROOT}/include", f"-I{host}",
           os.path.join(ROOT, "tests", "cpp", "host_sanity.cpp"), *srcs, "-o", exe]
    subprocess.run(cmd, check=True)
    work = tmp_path / "w"
    work.mkdir()
    r = subprocess.run([exe, os.path.join(ROOT, "scenes"), str(work)], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "host_sanity: clean" in r.stdout, r.stdout[-3000:]
    assert "runtime error" not in r.stdout and "AddressSanitizer" not in r.stdout, r.stdout[-3000:]


def test_oracle_under_asan_ubsan(tmp_path):
    from qaray_amd.host import load_scene_blob
    exe = str(tmp_path / "oracle_san")
    main = tmp_path / "main.c"
    main.write_text('''
#include <stdio.h>
#include <stdlib.h>
#include "qa_oracle.h"
int main(int argc, char **argv) {
  FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  void *blob = malloc(n); if (fread(blob, 1, n, f) != (size_t) n) return 3; fclose(f);
  int w = atoi(argv[2]), h = atoi(argv[3]);
  float *rgb = malloc(sizeof(float) * 3 * w * h), *d = malloc(sizeof(float) * w * h);
  uint32_t *ns = malloc(4 * w * h); qa_oracle_counters c;
  int rc = qa_oracle_render(blob, 0, 0, w, h, 2, 2, 5, 0x51A7A7, rgb, d, ns, 2, &c);
  double s = 0; for (int i = 0; i < 3 * w * h; ++i) s += rgb[i];
  printf("rc=%d sum=%.6f samples=%llu\\n", rc, s, (unsigned long long) c.samples);
  free(blob); free(rgb); free(d); free(ns); return rc;
}''')
    subprocess.run(["gcc", "-std=gnu11", *SAN, "-fopenmp", "-ffp-contract=off", f"-I{ROOT}/include", f"-I{ROOT}/oracle",
                    str(main), os.path.join(ROOT, "oracle", "qa_oracle.c"), "-o", exe, "-lm"], check=True)
    for scene, (w, h) in (("custom_textures.xml", (24, 18)), ("custom_softshadow.xml", (12, 9)), ("example_project12_box.xml", (24, 18))):
        blob = load_scene_blob(scene, size=(w, h))
        p = tmp_path / "scene.bin"
        blob.tofile(p)
        r = subprocess.run([exe, str(p), str(w), str(h)], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0 and "rc=0" in r.stdout, r.stdout[-3000:]
        assert "runtime error" not in r.stdout and "AddressSanitizer" not in r.stdout, r.stdout[-3000:]

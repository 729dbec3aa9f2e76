// Sanitizer driver for the host layer (built by tests/test_sanitizers.py with
// -fsanitize=address,undefined): loads every scene under scenes/, flattens it, exercises the
// image codecs and the framebuffer, and feeds the parsers malformed input.  Exit code 0 = clean.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "framebuffer.h"
#include "image.h"
#include "scene.h"
#include "xml.h"

using namespace qaray_hip;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("CHECK failed: %s (line %d)\n", #c, __LINE__); ++fails; } } while (0)

static void WriteFile(const std::string &p, const std::string &s) { std::ofstream(p, std::ios::binary) << s; }

int main(int argc, char **argv)
{
  if (argc < 3) { printf("usage: host_sanity <scenes dir> <tmp dir>\n"); return 2; }
  const std::string scenes = std::string(argv[1]) + "/", tmp = std::string(argv[2]) + "/";
  const char *names[] = {"example_project12_box.xml", "example_project3_sphere.xml", "example_project4.xml", "custom_textures.xml",
                         "custom_softshadow.xml", "trc_scene_xmas.xml", "trc_mtl_glass.xml"};
  for (const char *n : names) {
    Scene sc;
    sc.assetRoot = scenes;
    CHECK(LoadScene((scenes + n).c_str(), sc) == 1);
    const std::vector<unsigned char> blob = FlattenScene(sc);
    const qa_flat_header *h = reinterpret_cast<const qa_flat_header *>(blob.data());
    CHECK(h->magic == QA_FLAT_MAGIC && h->total_bytes == blob.size());
    CHECK(h->num_instances >= 1);
  }
  // tasking (src/tasking/parallel_for.h:59-95): every index exactly once, strides, thread-local storage, stop flag
  {
    tasking::set_num_of_threads(4);
    CHECK(tasking::get_num_of_threads() == 4);
    std::vector<int> seen(1000, 0);
    tasking::parallel_for(3, 1000, 7, [&](size_t i) { seen[i]++; });
    int bad = 0;
    for (size_t i = 0; i < seen.size(); ++i) bad += seen[i] != ((i >= 3 && (i - 3) % 7 == 0) ? 1 : 0);
    CHECK(bad == 0);
    tasking::ThreadLocalStorage<std::vector<int>> tls(std::vector<int>(1, 0));
    tasking::parallel_for(0, 64, 1, [&](size_t) { tls.local()[0]++; });
    CHECK(tls.local()[0] >= 0 && tls.data[0] == 0);
    tasking::parallel_for(5, 5, 1, [&](size_t) { ++fails; });      // empty range
    bool thrown = false;
    try { tasking::parallel_for(0, 100, 1, [&](size_t i) { if (i == 13) throw 13; }); } catch (int v) { thrown = v == 13; }
    CHECK(thrown);
    tasking::signal_stop();
    CHECK(tasking::has_stop_signal());
    tasking::signal_start();
    CHECK(!tasking::has_stop_signal());
  }
  // malformed XML never crashes
  const char *bad[] = {"", "<", "<xml", "<xml><scene></xml>", "<xml a=></xml>", "<xml a='1></xml>", "<!-- x", "<xml><a/><b></c></xml>",
                       "<xml><scene><object type=\"obj\" name=\"nope.obj\"><scale x=\"abc\"/></object></scene><camera/></xml>"};
  for (const char *b : bad) {
    XmlDocument d;
    (void) d.Parse(b);
    WriteFile(tmp + "bad.xml", b);
    Scene sc;
    sc.assetRoot = tmp;
    (void) LoadScene((tmp + "bad.xml").c_str(), sc);
  }
  // malformed OBJ / MTL
  const char *objs[] = {"v 1 2\nf 1 2 3\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n", "f 0 1 2\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 -2 -3\nf 1//1 2//1 3//1\n",
                        "mtllib nothere.mtl\nusemtl x\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 1 2 3 4\n", "v 1e400 -1e-400 nan\nvt\nvn 1\n"};
  for (const char *o : objs) {
    WriteFile(tmp + "t.obj", o);
    TriObj t;
    std::string err;
    if (t.Load((tmp + "t.obj").c_str(), true, &err)) {
      // whatever loaded must be internally consistent or be rejected later by the flattener's consumer
      CHECK(t.bvhElements.size() == t.NF());
    }
  }
  // PNG round trip + truncated / corrupted files
  {
    std::vector<unsigned char> img(37 * 21 * 3);
    for (size_t i = 0; i < img.size(); ++i) img[i] = (unsigned char) (i * 7 + 3);
    CHECK(SavePNG((tmp + "a.png").c_str(), img.data(), 37, 21, 3));
    int w = 0, h = 0;
    std::vector<unsigned char> back;
    CHECK(LoadPNG((tmp + "a.png").c_str(), w, h, back) && w == 37 && h == 21 && back == img);
    std::ifstream f(tmp + "a.png", std::ios::binary);
    std::string raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    for (size_t cut : {(size_t) 0, (size_t) 7, (size_t) 20, (size_t) 40, raw.size() / 2, raw.size() - 5}) {
      WriteFile(tmp + "cut.png", raw.substr(0, cut));
      (void) LoadPNG((tmp + "cut.png").c_str(), w, h, back);
    }
    for (size_t pos = 8; pos < raw.size(); pos += 97) {
      std::string c = raw;
      c[pos] = (char) (c[pos] ^ 0x5A);
      WriteFile(tmp + "cor.png", c);
      (void) LoadPNG((tmp + "cor.png").c_str(), w, h, back);
    }
    WriteFile(tmp + "p.ppm", "P6\n# c\n2 2\n255\nabcdefghijkl");
    CHECK(LoadPPM((tmp + "p.ppm").c_str(), w, h, back) && w == 2 && h == 2);
    WriteFile(tmp + "q.ppm", "P6\n99999 99999\n255\n");
    CHECK(!LoadPPM((tmp + "q.ppm").c_str(), w, h, back));
  }
  // framebuffer
  {
    FrameBuffer fb;
    fb.Init(16, 8);
    std::vector<float> rgb(16 * 8 * 3, 0.25f), z(16 * 8, 3.f);
    std::vector<uint32_t> ns(16 * 8, 4);
    z[5] = 1.0e30f;
    fb.Deposit(0, 0, 16, 8, rgb.data(), z.data(), ns.data(), 8, true);
    fb.ComputeZBufferImage();
    CHECK(fb.ComputeSampleCountImage() == 127);
    CHECK(fb.SaveImage((tmp + "c.png").c_str()) && fb.SaveZImage((tmp + "z.png").c_str()) && fb.SaveSampleCountImage((tmp + "s.png").c_str()));
    CHECK(fb.IsRenderDone());
  }
  printf(fails ? "host_sanity: %d failure(s)\n" : "host_sanity: clean\n", fails);
  return fails ? 1 : 0;
}

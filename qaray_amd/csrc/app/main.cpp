// qaray_hip — batch driver with the reference's command line (src/main.cpp:8-62):
//   qaray_hip [-batch] [-spp N] [-sppMin N] [-sppMax N] [-bounce N] [-srgb 0|1] [-threads N]
//             [-use-photon-map] [-photon-map-size N] [-caustics-map-size N] scene.xml
// plus what the reference has no flag for: -size W H, -seed S, -device D, -devices N (GPUs 0..N-1 of this node in one
// process: one host thread and one context per GPU, strips gathered on the first; Renderer::UseDevices), -out PREFIX, -root DIR,
// -photon-map-radius R, -caustics-map-radius R, -photon-map-bounce N, -caustics-map-bounce N.
// The reference's `-sppMax` sets sppMin by mistake (main.cpp:27-28); here it sets sppMax.
// Flow: Init -> LoadScene -> ComputeScene -> Render -> Terminate (main.cpp:55-59).
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "renderer.h"

using namespace qaray_hip;

int main(int argc, char **argv)
{
  setenv("GPU_MAX_HW_QUEUES", "8", 0);   // the staged integrator's tile groups each want a hardware queue (qa_wf.hip)
  RendererParam param;
  const char *file = nullptr;
  std::string out, root;
  int device = 0, w = -1, h = -1, devices = 0;
  if (argc < 2) { fprintf(stderr, "Error: insufficient input\n"); return -1; }
  for (int i = 1; i < argc; ++i) {
    const std::string s(argv[i]);
    auto next = [&]() { return (i + 1 < argc) ? argv[++i] : "0"; };
    if (s == "-batch") {}
    else if (s == "-spp") { const int t = atoi(next()); param.SetSPPMax(t); param.SetSPPMin(t); }
    else if (s == "-sppMin") param.SetSPPMin(atoi(next()));
    else if (s == "-sppMax") param.SetSPPMax(atoi(next()));
    else if (s == "-bounce") Material::maxBounce = atoi(next());
    else if (s == "-srgb") param.SetSRGBFlag(atoi(next()) != 0);
    else if (s == "-threads") (void) next();  // CPU thread count has no meaning here
    else if (s == "-use-photon-map") param.SetPhotonMappingFlag(true);
    else if (s == "-photon-map-size") param.SetPhotonMapSize((size_t) atoi(next()));
    else if (s == "-caustics-map-size") param.SetCausticsMapSize((size_t) atoi(next()));
    else if (s == "-photon-map-radius") param.SetPhotonMapRadius((float) atof(next()));
    else if (s == "-caustics-map-radius") param.SetCausticsMapRadius((float) atof(next()));
    else if (s == "-photon-map-bounce") param.SetPhotonMapBounce((size_t) atoi(next()));
    else if (s == "-caustics-map-bounce") param.SetCausticsMapBounce((size_t) atoi(next()));
    else if (s == "-size") { w = atoi(next()); h = atoi(next()); }
    else if (s == "-seed") param.seed = (uint32_t) strtoul(next(), nullptr, 0);
    else if (s == "-device") device = atoi(next());
    else if (s == "-devices") devices = atoi(next());
    else if (s == "-out") out = next();
    else if (s == "-root") root = next();
    else file = argv[i];
  }
  if (!file) { fprintf(stderr, "Error: no scene file\n"); return -1; }
  try {
    Renderer renderer(param, device);
    if (devices > 0) {
      std::vector<int> ids;
      for (int d = 0; d < devices; ++d) ids.push_back(device + d);
      renderer.UseDevices(ids);
    }
    renderer.outputPrefix = out;
    renderer.Init();
    Scene scene;
    scene.assetRoot = root;
    if (!scene.assetRoot.empty() && scene.assetRoot.back() != '/') scene.assetRoot += '/';
    LoadSceneInSilentMode(false);
    if (!LoadScene(file, scene)) { fprintf(stderr, "Failed to load %s\n", file); return 1; }
    if (w > 0 && h > 0) { scene.camera.imgWidth = w; scene.camera.imgHeight = h; }
    renderer.ComputeScene(renderImage, scene);
    renderer.Render();
    const qa_counters &c = renderer.Counters();
    printf("samples %llu  casts %llu + %llu shadow  %.3f Msamples/s\n", (unsigned long long) c.samples,
           (unsigned long long) c.casts_normal, (unsigned long long) c.casts_shadow, c.samples / renderer.LastSeconds() * 1e-6);
    renderer.KillTimer();
    renderer.Terminate();
  } catch (const std::exception &e) {
    fprintf(stderr, "qaray_hip: %s\n", e.what());
    return 2;
  }
  return 0;
}

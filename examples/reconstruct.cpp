/*
 * PLY files of splats in, PLY mesh out, everything between on the device(s): the shape of the reference's main pipeline
 * (src/mlsgpu_core.cpp:560-760: load -> bounding grid -> Bucket::bucket -> BucketLoader -> device workers -> mesher ->
 * FastPly::Writer) with this repository's pieces:
 *   SplatSet::FileSet    mlsgpu::hip::FileSet         (several files, reader threads, bounded pinned memory, H2D overlap)
 *   bounding grid        mlsgpu_hip_bounding_grid     (device reduction)
 *   Bucket::bucket       mlsgpu::hip::Bucket::bucket  (device)
 *   BucketLoader + CopyGroup + DeviceWorkerGroup      mlsgpu::hip::BucketFarm::submitDevice (device gather + transform
 *                        into a device item of the least-loaded GPU -- a peer copy when that is another GPU -- then four
 *                        worker threads per GPU: octree, MLS, marching, scale/bias)
 *   OOCMesher            --weld device: mlsgpu::hip::DeviceMesher (ship-outs appended in one GPU's HBM, weld / components /
 *                        prune there); --weld host: every ship-out read back through the pinned circular buffer and welded
 *                        by mlsgpu::hip::OOCMesher on the mesher thread (the reference's route)
 *
 * usage: reconstruct [--devices 0,1,...] [--weld device|host] [--tmp-dir DIR] [--buffer BYTES] <in.ply> [more.ply ...] <out.ply>
 *                    <spacing> [smooth=4] [levels=6] [subsampling=3] [prune=0.02] [maxSplats=2097152]
 * (defaults as src/mlsgpu_core.cpp:86-135: --fit-smooth 4, --levels 6, --subsampling 3, --fit-prune 0.02)
 */
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <sstream>
#include <vector>

#include "../mlsgpu_amd/host/mlsgpu_hip.hpp"

using namespace mlsgpu::hip;

static bool isPly(const std::string &a) { return a.size() > 4 && a.compare(a.size() - 4, 4, ".ply") == 0; }

int main(int argc, char **argv)
{
    std::vector<std::int32_t> devices(1, 0);
    bool hostWeld = false;
    std::uint64_t bufferBytes = 0, hbmSplats = 0;
    std::string tmpDir;
    std::vector<std::string> plys, rest;
    for (int i = 1; i < argc; i++)
    {
        const std::string a = argv[i];
        if (a == "--devices" && i + 1 < argc)
        {
            devices.clear();
            std::istringstream ss(argv[++i]);
            std::string tok;
            while (std::getline(ss, tok, ','))
                devices.push_back(atoi(tok.c_str()));
        }
        else if (a == "--weld" && i + 1 < argc)
            hostWeld = std::string(argv[++i]) == "host";
        else if (a == "--buffer" && i + 1 < argc)
            bufferBytes = strtoull(argv[++i], NULL, 10);
        else if (a == "--tmp-dir" && i + 1 < argc)             // --weld host: the welder's blocks in temporary files there (src/mlsgpu_core.cpp --tmp-dir)
            tmpDir = argv[++i];
        else if (a == "--hbm-splats" && i + 1 < argc)       // the cloud does not fit the device: stream it, this many at a time
            hbmSplats = strtoull(argv[++i], NULL, 10);
        else if (rest.empty() && isPly(a))
            plys.push_back(a);
        else
            rest.push_back(a);
    }
    if (plys.size() < 2 || rest.empty() || devices.empty())
    {
        std::cerr << "usage: reconstruct [--devices 0,1] [--weld device|host] [--tmp-dir DIR] [--buffer BYTES] [--hbm-splats N] in.ply [more.ply ...] out.ply "
                     "spacing [smooth] [levels] [subsampling] [prune] [maxSplats]\n";
        return 2;
    }
    const std::string outName = plys.back();
    plys.pop_back();
    const float spacing = (float) atof(rest[0].c_str());
    const float smooth = rest.size() > 1 ? (float) atof(rest[1].c_str()) : 4.0f;
    const unsigned levels = rest.size() > 2 ? (unsigned) atoi(rest[2].c_str()) : 6;
    const unsigned subsampling = rest.size() > 3 ? (unsigned) atoi(rest[3].c_str()) : 3;
    const double prune = rest.size() > 4 ? atof(rest[4].c_str()) : 0.02;
    const std::uint64_t maxSplats = rest.size() > 5 ? strtoull(rest[5].c_str(), NULL, 10) : 2097152;
    const std::uint32_t maxCells = (1u << (levels + subsampling - 1)) - 1;        // src/mlsgpu_core.cpp:672-673
    const std::uint32_t microCells = std::min<std::uint32_t>(63, maxCells);          // --leaf-cells 63, :113,674
    try
    {
        FileSet files(smooth, std::numeric_limits<float>::infinity());
        for (const std::string &p : plys)
            files.addFile(p);
        if (bufferBytes)
            files.setBufferSize(bufferBytes);
        const std::uint64_t numSplats = files.maxSplats();
        if (numSplats == 0)
        {
            std::cerr << "no splats\n";
            return 1;
        }
        const int home = devices[0];                            // the cloud (and the device sink) live on the first GPU
        Context ctx(home);
        const bool streamed = hbmSplats != 0;                   // the set is larger than what may be resident at once
        const std::uint64_t chunkSplats = streamed ? std::max<std::uint64_t>(hbmSplats / 4, 1) : 0;
        Buffer<Splat> cloud(ctx, streamed ? 1 : numSplats);
        Bucket::Grid grid;
        if (streamed)
            grid = files.boundingGrid(ctx, spacing, microCells, chunkSplats);      // one pass over the files
        else
        {
            // files -> HBM: reader threads decode into pinned quarters while earlier chunks travel; no host copy of the cloud
            files.load(ctx, cloud, 0, numSplats);
            check(mlsgpu_hip_bounding_grid(ctx.get(), cloud.get(), numSplats, spacing, microCells, &grid));
        }

        mlsgpu_worker_config cfg;
        std::memset(&cfg, 0, sizeof(cfg));
        cfg.maxBucketSplats = maxSplats;
        cfg.maxCells = maxCells;
        cfg.levels = levels;
        cfg.subsampling = subsampling;
        cfg.boundaryLimit = 1.0f;
        cfg.shape = MLSGPU_SHAPE_SPHERE;
        cfg.gridSpacing = spacing;                             // ScaleBiasFilter: vertex * spacing + grid.getVertex(0,0,0)
        for (int i = 0; i < 3; i++)
            cfg.gridOrigin[i] = grid.reference[i] + spacing * (float) grid.extents[2 * i];

        DeviceMesher deviceMesher(ctx);
        deviceMesher.setPruneThreshold(prune);
        OOCMesher hostMesher;
        hostMesher.setPruneThreshold(prune);
        if (!tmpDir.empty())
            hostMesher.setTmpDir(tmpDir, 0);
        std::size_t bins = 0, written = 0;
        std::uint64_t st[8];
        {
            BucketFarm farm(devices, cfg, 4 /* --device-threads */, 1, hostWeld ? NULL : &deviceMesher);
            if (hostWeld)
                farm.setHostOutput(std::uint64_t(512) << 20 /* --mem-mesh */, hostMesher);
            if (streamed)
                // out of core: the files streamed through a chunk buffer, a batch of top-level regions resident at a time
                Bucket::bucketStream(ctx, files, grid, maxSplats, maxCells, 0, microCells, std::uint64_t(1) << 30, hbmSplats,
                                     chunkSplats, 0, [&](const Bucket::Bin &bin)
                {
                    farm.submitDevice(home, bin, grid, 0);
                    bins++;
                });
            else
                Bucket::bucket(ctx, cloud, numSplats, grid, maxSplats, maxCells, 0, microCells, std::uint64_t(1) << 30,
                               [&](const Bucket::Bin &bin)
                {
                    farm.submitDevice(home, cloud, bin, grid, 0);
                    bins++;
                });
            farm.finish();
        }
        const std::vector<std::string> comments(1, "mlsgpu-hip example: reconstruct");
        if (hostWeld)
        {
            written = hostMesher.write([&](std::uint64_t) { return outName; }, comments);
            hostMesher.getStatistics(st);
        }
        else
        {
            written = deviceMesher.write([&](std::uint64_t) { return outName; }, comments);
            deviceMesher.getStatistics(st);
        }
        std::printf("files in %zu splats %llu grid %d..%d %d..%d %d..%d bins %zu devices %zu weld %s files %zu vertices %llu "
                    "triangles %llu components %llu kept %llu\n",
                    plys.size(), (unsigned long long) numSplats, grid.extents[0], grid.extents[1], grid.extents[2],
                    grid.extents[3], grid.extents[4], grid.extents[5], bins, devices.size(), hostWeld ? "host" : "device",
                    written, (unsigned long long) st[4], (unsigned long long) st[5], (unsigned long long) st[2],
                    (unsigned long long) st[3]);
    }
    catch (std::exception &e)
    {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}

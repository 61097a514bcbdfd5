/*
 * PLY of splats in, PLY mesh out, everything between on the device: the in-core shape of the reference's main
 * pipeline (src/mlsgpu_core.cpp:560-760: load -> bounding grid -> Bucket::bucket -> BucketLoader -> device worker ->
 * mesher -> FastPly::Writer) with this repository's pieces:
 *   FastPly::Reader      mlsgpu_hip_ply_open / _load (host threads decode while batches travel to the device)
 *   bounding grid        mlsgpu_hip_bounding_grid    (device reduction)
 *   Bucket::bucket       mlsgpu::hip::Bucket::bucket (device)
 *   BucketLoader + CopyGroup + DeviceWorkerGroup   mlsgpu_hip_farm_submit_device (device gather + transform into a
 *                        device item, then four worker threads: octree, MLS, marching, scale/bias)
 *   OOCMesher            mlsgpu::hip::DeviceMesher   (device weld / components / prune), FastPly::Writer on the host
 *
 * usage: reconstruct <in.ply> <out.ply> <spacing> [smooth=4] [levels=6] [subsampling=3] [prune=0.02] [maxSplats=2097152]
 * (defaults as src/mlsgpu_core.cpp:86-135: --fit-smooth 4, --levels 6, --subsampling 3, --fit-prune 0.02)
 */
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <vector>

#include "../mlsgpu_amd/host/mlsgpu_hip.hpp"

using namespace mlsgpu::hip;

int main(int argc, char **argv)
{
    if (argc < 4)
    {
        std::cerr << "usage: reconstruct in.ply out.ply spacing [smooth] [levels] [subsampling] [prune] [maxSplats]\n";
        return 2;
    }
    const float spacing = (float) atof(argv[3]);
    const float smooth = argc > 4 ? (float) atof(argv[4]) : 4.0f;
    const unsigned levels = argc > 5 ? (unsigned) atoi(argv[5]) : 6;
    const unsigned subsampling = argc > 6 ? (unsigned) atoi(argv[6]) : 3;
    const double prune = argc > 7 ? atof(argv[7]) : 0.02;
    const std::uint64_t maxSplats = argc > 8 ? strtoull(argv[8], NULL, 10) : 2097152;
    const std::uint32_t maxCells = (1u << (levels + subsampling - 1)) - 1;        // src/mlsgpu_core.cpp:672-673
    const std::uint32_t microCells = std::min<std::uint32_t>(63, maxCells);          // --leaf-cells 63, :113,674
    try
    {
        mlsgpu_ply_reader *reader = NULL;
        check(mlsgpu_hip_ply_open(argv[1], smooth, std::numeric_limits<float>::infinity(), &reader));
        const std::uint64_t numSplats = mlsgpu_hip_ply_size(reader);
        if (numSplats == 0)
        {
            std::cerr << "no splats\n";
            mlsgpu_hip_ply_close(reader);
            return 1;
        }
        Context ctx(0);
        Buffer<Splat> cloud(ctx, numSplats);
        // file -> HBM: threaded decode overlapped with the host-to-device copies, no host copy of the whole cloud
        const int loaded = mlsgpu_hip_ply_load(reader, ctx.get(), 0, numSplats, cloud.get(), 0);
        mlsgpu_hip_ply_close(reader);
        check(loaded);
        Bucket::Grid grid;
        check(mlsgpu_hip_bounding_grid(ctx.get(), cloud.get(), numSplats, spacing, microCells, &grid));

        mlsgpu_farm_config fcfg;
        std::memset(&fcfg, 0, sizeof(fcfg));
        fcfg.numDevices = 1;
        fcfg.workersPerDevice = 4;                              // --device-threads
        fcfg.spare = 1;
        mlsgpu_worker_config &cfg = fcfg.worker;
        cfg.maxBucketSplats = maxSplats;
        cfg.maxCells = maxCells;
        cfg.levels = levels;
        cfg.subsampling = subsampling;
        cfg.boundaryLimit = 1.0f;
        cfg.shape = MLSGPU_SHAPE_SPHERE;
        cfg.gridSpacing = spacing;                             // ScaleBiasFilter: vertex * spacing + grid.getVertex(0,0,0)
        for (int i = 0; i < 3; i++)
            cfg.gridOrigin[i] = grid.reference[i] + spacing * (float) grid.extents[2 * i];

        DeviceMesher mesher(ctx);
        mesher.setPruneThreshold(prune);
        // the farm's output functor (OutputGenerator of src/workers.h:225): every ship-out goes to the device mesher
        struct Sink
        {
            static int call(void *user, int, std::uint64_t, mlsgpu_ctx *workerCtx, const mlsgpu_mesh *mesh)
            {
                return mlsgpu_hip_mesher_add(static_cast<DeviceMesher *>(user)->get(), workerCtx, 0, mesh);
            }
        };
        mlsgpu_farm *farm = NULL;
        check(mlsgpu_hip_farm_create(&fcfg, &Sink::call, &mesher, &farm));
        std::size_t bins = 0;
        try
        {
            Bucket::bucket(ctx, cloud, numSplats, grid, maxSplats, maxCells, 0, microCells, std::uint64_t(1) << 30,
                           [&](const Bucket::Bin &bin)
            {
                std::int32_t low[3];
                std::uint32_t nv[3];
                for (int i = 0; i < 3; i++)
                {
                    low[i] = bin.extents[2 * i] - grid.extents[2 * i];
                    nv[i] = (std::uint32_t) (bin.extents[2 * i + 1] - bin.extents[2 * i] + 1);
                }
                check(mlsgpu_hip_farm_submit_device(farm, 0, cloud.get(), bin.dIds, bin.numSplats, &grid, low, nv, bins));
                bins++;
            });
            check(mlsgpu_hip_farm_finish(farm));
        }
        catch (...)
        {
            mlsgpu_hip_farm_destroy(farm);
            throw;
        }
        mlsgpu_hip_farm_destroy(farm);
        const std::string outName = argv[2];
        const std::size_t files = mesher.write([&](std::uint64_t) { return outName; },
                                               std::vector<std::string>(1, "mlsgpu-hip example: reconstruct"));
        std::uint64_t st[8];
        mesher.getStatistics(st);
        std::printf("splats %llu grid %d..%d %d..%d %d..%d bins %zu files %zu vertices %llu triangles %llu components %llu kept %llu\n",
                    (unsigned long long) numSplats, grid.extents[0], grid.extents[1], grid.extents[2], grid.extents[3], grid.extents[4],
                    grid.extents[5], bins, files, (unsigned long long) st[4], (unsigned long long) st[5],
                    (unsigned long long) st[2], (unsigned long long) st[3]);
    }
    catch (std::exception &e)
    {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}

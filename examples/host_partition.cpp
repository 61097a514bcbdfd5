/*
 * Bucket::bucket on the device from C++ (mlsgpu::hip::Bucket of mlsgpu_amd/host/mlsgpu_hip.hpp): uploads a
 * world-space cloud, partitions it with the reference's arguments and prints every bin -- the host side of
 * src/mlsgpu_core.cpp:655-678 (doBucket) with a processor that just reports.
 *
 * usage: host_partition <splats.bin> refX refY refZ spacing x0 x1 y0 y1 z0 z1 maxSplats maxCells chunkCells microCells maxSplit
 * prints: one line per bin "x0 x1 y0 y1 z0 z1 chunkX chunkY chunkZ depth numSplats sumOfIds", then "bins N"
 */
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "../mlsgpu_amd/host/mlsgpu_hip.hpp"

using namespace mlsgpu::hip;

int main(int argc, char **argv)
{
    if (argc != 17)
    {
        std::cerr << "usage: host_partition splats.bin refX refY refZ spacing x0 x1 y0 y1 z0 z1 maxSplats maxCells "
                     "chunkCells microCells maxSplit\n";
        return 2;
    }
    std::vector<Splat> splats;
    {
        std::ifstream in(argv[1], std::ios::binary | std::ios::ate);
        splats.resize((std::size_t) in.tellg() / sizeof(Splat));
        in.seekg(0);
        in.read(reinterpret_cast<char *>(splats.data()), splats.size() * sizeof(Splat));
    }
    Bucket::Grid grid;
    for (int i = 0; i < 3; i++)
        grid.reference[i] = (float) atof(argv[2 + i]);
    grid.spacing = (float) atof(argv[5]);
    for (int i = 0; i < 6; i++)
        grid.extents[i] = atoi(argv[6 + i]);
    try
    {
        Context ctx(0);
        Buffer<Splat> dSplats(ctx, splats.size() > 0 ? splats.size() : 1);
        if (!splats.empty())
            dSplats.write(splats.data(), splats.size());
        std::size_t bins = 0;
        Bucket::bucket(ctx, dSplats, splats.size(), grid, strtoull(argv[12], NULL, 10), (std::uint32_t) atoi(argv[13]),
                       (std::uint32_t) atoi(argv[14]), (std::uint32_t) atoi(argv[15]), strtoull(argv[16], NULL, 10),
                       [&](const Bucket::Bin &bin)
        {
            std::vector<std::uint32_t> ids(bin.numSplats);
            check(mlsgpu_hip_memcpy_d2h(ctx.get(), ids.data(), bin.dIds, ids.size() * sizeof(std::uint32_t), 0));
            unsigned long long sum = 0;
            for (std::uint32_t id : ids)
                sum += id;
            std::printf("%d %d %d %d %d %d %llu %llu %llu %u %llu %llu\n", bin.extents[0], bin.extents[1], bin.extents[2],
                        bin.extents[3], bin.extents[4], bin.extents[5], (unsigned long long) bin.chunk[0],
                        (unsigned long long) bin.chunk[1], (unsigned long long) bin.chunk[2], bin.depth,
                        (unsigned long long) bin.numSplats, sum);
            bins++;
        });
        std::printf("bins %zu\n", bins);
    }
    catch (Bucket::DensityError &e)
    {
        std::printf("density %llu\n", (unsigned long long) e.getCellSplats());
    }
    return 0;
}

// tools/pcl_golden/dump_pcl_golden.cpp — for a maintainer who has PCL (1.7 - 1.9, the versions the reference's README names); NOT built here
// (this container and the GPU box have no PCL) and not part of the product or of the default tests.
//
//   g++ -O2 -std=c++14 dump_pcl_golden.cpp -o dump_pcl_golden $(pkg-config --cflags --libs pcl_filters pcl_segmentation pcl_features pcl_common)
//   python tools/pcl_golden/make_inputs.py pcl_in.bin
//   ./dump_pcl_golden pcl_in.bin pcl_out.bin
//   python tools/pcl_golden/make_inputs.py --pack pcl_out.bin tests/golden/pcl_golden.npz
//
// It runs the three PCL calls the reference makes on the plane path, exactly as the reference configures them, on the inputs of make_inputs.py:
//   VOX   pcl::VoxelGrid<PointXYZRGB>(0.1)                                   src/Frame.cc:672-678
//   SAC   pcl::SACSegmentation<PointXYZRGB> (plane, RANSAC, optimised, th)   src/Frame.cc:774-786
//   NRM   pcl::IntegralImageNormalEstimation (AVERAGE_3D_GRADIENT, 0.05, 10) src/Frame.cc:715-726
// tests/test_oracle_pcl_golden.py compares oracle/planepost_oracle.cpp and oracle/normals_oracle.cpp with the result when the .npz exists; that
// is what would retire their "PARITY UNPINNED" headers.
#include <pcl/ModelCoefficients.h>
#include <pcl/features/integral_image_normal.h>
#include <pcl/filters/voxel_grid.h>
#include <pcl/point_types.h>
#include <pcl/segmentation/sac_segmentation.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef pcl::PointXYZRGB PointT;
typedef pcl::PointCloud<PointT> Cloud;

static bool rd(FILE* f, void* p, size_t n) { return std::fread(p, 1, n, f) == n; }

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    FILE* fi = std::fopen(argv[1], "rb"); FILE* fo = std::fopen(argv[2], "wb");
    if (!fi || !fo) return 2;
    int32_t ncase;
    if (!rd(fi, &ncase, 4)) return 2;
    std::fwrite(&ncase, 4, 1, fo);
    for (int c = 0; c < ncase; c++) {
        int32_t kind, n, w, h; double th;
        if (!rd(fi, &kind, 4) || !rd(fi, &n, 4) || !rd(fi, &w, 4) || !rd(fi, &h, 4) || !rd(fi, &th, 8)) return 2;
        std::vector<float> xyz((size_t)n * 3);
        if (!rd(fi, xyz.data(), xyz.size() * 4)) return 2;
        Cloud::Ptr cloud(new Cloud());
        for (int i = 0; i < n; i++) { PointT p; p.x = xyz[i * 3]; p.y = xyz[i * 3 + 1]; p.z = xyz[i * 3 + 2]; cloud->points.push_back(p); }
        std::fwrite(&kind, 4, 1, fo);
        if (kind == 0) {                       // VOX
            pcl::VoxelGrid<PointT> voxel;
            voxel.setLeafSize(0.1, 0.1, 0.1);
            Cloud::Ptr coarse(new Cloud());
            voxel.setInputCloud(cloud);
            voxel.filter(*coarse);
            const int32_t m = (int32_t)coarse->points.size();
            std::fwrite(&m, 4, 1, fo);
            for (auto& p : coarse->points) { const float v[3] = {p.x, p.y, p.z}; std::fwrite(v, 4, 3, fo); }
        } else if (kind == 1) {                // SAC
            pcl::ModelCoefficients::Ptr coefficients(new pcl::ModelCoefficients);
            pcl::PointIndices::Ptr inliers(new pcl::PointIndices);
            pcl::SACSegmentation<PointT> seg;
            seg.setOptimizeCoefficients(true);
            seg.setModelType(pcl::SACMODEL_PLANE);
            seg.setMethodType(pcl::SAC_RANSAC);
            seg.setDistanceThreshold(th);
            seg.setInputCloud(cloud);
            seg.segment(*inliers, *coefficients);
            const int32_t ni = (int32_t)inliers->indices.size(), nc = (int32_t)coefficients->values.size();
            std::fwrite(&ni, 4, 1, fo); std::fwrite(&nc, 4, 1, fo);
            std::fwrite(coefficients->values.data(), 4, nc, fo);
        } else {                               // NRM: organised cloud w x h
            cloud->width = w; cloud->height = h;
            pcl::IntegralImageNormalEstimation<PointT, pcl::Normal> ne;
            pcl::PointCloud<pcl::Normal>::Ptr normals(new pcl::PointCloud<pcl::Normal>);
            ne.setNormalEstimationMethod(ne.AVERAGE_3D_GRADIENT);
            ne.setMaxDepthChangeFactor(0.05f);
            ne.setNormalSmoothingSize(10.0f);
            ne.setInputCloud(cloud);
            ne.compute(*normals);
            const int32_t m = (int32_t)normals->points.size();
            std::fwrite(&m, 4, 1, fo);
            for (auto& q : normals->points) { const float v[3] = {q.normal_x, q.normal_y, q.normal_z}; std::fwrite(v, 4, 3, fo); }
        }
    }
    std::fclose(fi); std::fclose(fo);
    return 0;
}

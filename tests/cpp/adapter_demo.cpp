// Exercises include/loam_livox_adapter.hpp the way the reference's node shells call Livox_laser /
// Point_cloud_registration (laser_feature_extractor.hpp:285-330, laser_mapping.hpp:1405), with a stand-in for
// pcl::PointCloud<pcl::PointXYZI>.  argv: <scan.bin> <corner_map.bin> <surf_map.bin> <pose_init.bin> <out.txt>
// (.bin = raw float32 xyzi rows / 7 float64) [line_res plane_res]: with the two optional leaf sizes the features are
// voxel-filtered before registration.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "../../include/loam_livox_adapter.hpp"

struct PointXYZI {
    float x, y, z, intensity;
};
struct Cloud {
    std::vector<PointXYZI> points;
    size_t size() const { return points.size(); }
};
struct KdTreeStub {};

static Cloud load_cloud(const char *path)
{
    Cloud c;
    FILE *f = fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    PointXYZI p;
    while (fread(&p, sizeof(p), 1, f) == 1) c.points.push_back(p);
    fclose(f);
    return c;
}

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    try {
        Cloud scan = load_cloud(argv[1]);
        auto map_corner = std::make_shared<Cloud>(load_cloud(argv[2]));
        auto map_surf = std::make_shared<Cloud>(load_cloud(argv[3]));
        double pose[7];
        FILE *f = fopen(argv[4], "rb");
        if (!f || fread(pose, sizeof(double), 7, f) != 7) return 3;
        fclose(f);

        loam_livox_hip::Livox_laser m_livox;
        m_livox.thr_corner_curvature = 0.05f;  // laser_feature_extractor.hpp:152-154
        m_livox.thr_surface_curvature = 0.01f;
        m_livox.minimum_view_angle = 10;
        m_livox.m_livox_min_allow_dis = 0.1f;  // :854
        m_livox.m_livox_min_sigma = 7e-4f;     // :859
        m_livox.piecewise_number = 1;
        m_livox.max_points = 30000;
        std::vector<Cloud> laserCloudScans = m_livox.extract_laser_features(scan, 5.0);
        if (laserCloudScans.size() <= 5) return 4;  // :287
        const int m_laser_scan_number = (int)laserCloudScans.size();
        const int end_idx = (int)laserCloudScans[m_laser_scan_number - 1].size() - 1;
        const float piece_start = ((float)m_livox.find_pt_info(laserCloudScans[0].points[0])->idx) / m_livox.m_pts_info_vec.size();
        const float piece_end =
            ((float)m_livox.find_pt_info(laserCloudScans[m_laser_scan_number - 1].points[end_idx])->idx) / m_livox.m_pts_info_vec.size();
        auto corners = std::make_shared<Cloud>(), surface = std::make_shared<Cloud>();
        Cloud full;
        m_livox.get_features(*corners, *surface, full, piece_start, piece_end);

        // m_if_input_downsample_mode (laser_mapping.hpp:742-743,1367-1373): argv[6] = line_res, argv[7] = plane_res
        size_t n_corner_raw = corners->size(), n_surf_raw = surface->size();
        if (argc >= 8) {
            loam_livox_hip::VoxelGrid<Cloud> down_sample_filter_corner, down_sample_filter_surface;
            const float line_res = (float)atof(argv[6]), plane_res = (float)atof(argv[7]);
            down_sample_filter_corner.setLeafSize(line_res, line_res, line_res);
            down_sample_filter_surface.setLeafSize(plane_res, plane_res, plane_res);
            auto corner_stack = std::make_shared<Cloud>(), surf_stack = std::make_shared<Cloud>();
            down_sample_filter_corner.setInputCloud(corners);
            down_sample_filter_corner.filter(*corner_stack);
            down_sample_filter_surface.setInputCloud(surface);
            down_sample_filter_surface.filter(*surface);  // in place, like laser_feature_extractor.hpp:372-373
            corners = corner_stack;
        }

        loam_livox_hip::Point_cloud_registration pc_reg;
        pc_reg.m_current_frame_index = 100;
        pc_reg.m_mapping_init_accumulate_frames = 50;
        pc_reg.m_para_icp_max_iterations = 5;
        pc_reg.m_para_cere_max_iterations = 20;
        pc_reg.m_para_max_angular_rate = 20.0f;
        pc_reg.m_para_max_speed = 0.3f;
        pc_reg.max_features = 30000;
        // init_pointcloud_registration, laser_mapping.hpp:1290-1294: the node assigns the pose members
        pc_reg.m_q_w_curr.x() = pose[0], pc_reg.m_q_w_curr.y() = pose[1], pc_reg.m_q_w_curr.z() = pose[2], pc_reg.m_q_w_curr.w() = pose[3];
        for (int i = 0; i < 3; i++) pc_reg.m_t_w_curr(i) = pose[4 + i];
        pc_reg.m_q_w_last = pc_reg.m_q_w_curr;
        pc_reg.m_t_w_last = pc_reg.m_t_w_curr;
        KdTreeStub kd_c, kd_s;
        const int reg_res = pc_reg.find_out_incremental_transfrom(map_corner, map_surf, kd_c, kd_s, corners, surface);
        Cloud world;
        pc_reg.pointcloudAssociateToMap(*corners, world);
        // the per-point form the node uses in "Add new frame" (laser_mapping.hpp:1424, 1430) must agree with the cloud form
        for (size_t i = 0; i < corners->size(); i++) {
            PointXYZI sel;
            pc_reg.pointAssociateToMap(&corners->points[i], &sel, 1.0, 0);
            if (sel.x != world.points[i].x || sel.y != world.points[i].y || sel.z != world.points[i].z) return 6;
        }
        if (pc_reg.m_q_w_incre.w() != pc_reg.m_para_buffer_incremental[3] || pc_reg.m_t_w_incre(1) != pc_reg.m_para_buffer_incremental[5]) return 7;
        if (reg_res == 1 && pc_reg.m_final_opt_summary.BriefReport().empty()) return 8;

        // "Add new frame" + match-buffer refresh through the adapter (laser_mapping.hpp:1417-1478, 517-546)
        size_t n_hist_corner = 0, n_hist_surf = 0;
        int64_t n_cell_corner = 0, n_cell_surf = 0, n_cells = 0, n_cell_pts = 0;
        if (reg_res) {
            loam_livox_hip::History_buffer history(4, 30000, 0.1f, 0.4f);
            history.enable_cell_map(1 << 16, 1.0f, 5000);
            const bool pushed = history.add(*corners, *surface, pc_reg.m_para_buffer_RT);
            ll_map *scratch_map = nullptr;
            loam_livox_hip::check(ll_map_create(0, &scratch_map), "ll_map_create");
            int64_t nc = 0, ns = 0;
            history.refresh(scratch_map, &nc, &ns);
            Cloud buf_c, buf_s;
            history.map_cloud(0, buf_c);
            history.map_cloud(1, buf_s);
            // the same frame through the cell maps (m_matching_mode == 1, laser_mapping.hpp:471-546)
            history.refresh_cells(scratch_map, pc_reg.m_para_buffer_RT, 100.0f, 100.0f, 45.0f, 1, &n_cell_corner, &n_cell_surf);
            history.cell_map_size(1, &n_cells, &n_cell_pts);
            ll_map_destroy(scratch_map);
            if (!pushed || (int64_t)buf_c.size() != nc || (int64_t)buf_s.size() != ns) return 5;
            n_hist_corner = buf_c.size();
            n_hist_surf = buf_s.size();
        }

        // the scan's full cloud through a cell map: labels, key-frame images, self-similarity (cell_map_keyframe.hpp)
        int64_t n_map_cells = 0;
        int n_line_cells = 0, n_plane_cells = 0;
        float self_sim = 0.f, ratio_nz[4] = {0, 0, 0, 0};
        size_t n_cell_vec_first = 0, n_cell_vec_second = 0;
        {
            loam_livox_hip::Points_cloud_map cell_map(1 << 16, 1.0f);
            std::set<loam_livox_hip::Points_cloud_map::Cell_index> cell_vec;
            cell_map.append_cloud(full, &cell_vec);  // append_cloud( pts, &cell_vec ): every cell of the first cloud ...
            n_cell_vec_first = cell_vec.size();
            n_map_cells = cell_map.get_cells_size();
            std::vector<int32_t> ftype;
            std::vector<float> fvec;
            cell_map.determine_features(ftype, fvec);
            for (int32_t t : ftype) {
                n_line_cells += t == loam_livox_hip::Points_cloud_map::e_feature_line;
                n_plane_cells += t == loam_livox_hip::Points_cloud_map::e_feature_plane;
            }
            std::vector<float> images;
            cell_map.analyze(images, ratio_nz);
            self_sim = loam_livox_hip::Points_cloud_map::max_similiarity_of_two_image(images.data() + 3600, images.data() + 3600);
            cell_map.append_cloud(full, &cell_vec);  // ... afterwards the cells that received at least three points
            n_cell_vec_second = cell_vec.size();
        }

        FILE *o = fopen(argv[5], "w");
        fprintf(o, "%d %zu %zu %zu %d %zu %zu\n", m_laser_scan_number, n_corner_raw, n_surf_raw, full.size(), reg_res, corners->size(), surface->size());
        for (int i = 0; i < 7; i++) fprintf(o, "%.17g ", pc_reg.m_para_buffer_RT[i]);
        fprintf(o, "\n%.9g %.9g\n", piece_start, piece_end);
        fprintf(o, "%zu %zu\n", n_hist_corner, n_hist_surf);
        fprintf(o, "%lld %lld %lld %lld\n", (long long)n_cell_corner, (long long)n_cell_surf, (long long)n_cells, (long long)n_cell_pts);
        fprintf(o, "%lld %d %d %.9g %.9g %.9g %zu %zu\n", (long long)n_map_cells, n_line_cells, n_plane_cells, self_sim, ratio_nz[0], ratio_nz[1],
                n_cell_vec_first, n_cell_vec_second);
        fclose(o);
    } catch (const std::exception &e) {
        fprintf(stderr, "adapter_demo: %s\n", e.what());
        return 1;
    }
    return 0;
}

// loam_livox_adapter.hpp -- header-only C++ adapter that gives the reference's node shells the call surface they
// already use (hku-mars/loam_livox), backed by the C ABI of libloamlivox_hip.so:
//
//   loam_livox_hip::Livox_laser               <->  class Livox_laser               source/livox_feature_extractor.hpp:77
//   loam_livox_hip::Point_cloud_registration  <->  class Point_cloud_registration  source/point_cloud_registration.hpp:38
//
// The adapter is templated on the cloud type so that it compiles with or without PCL: any type with a
// `points` std::vector whose elements have float members x, y, z, intensity works (pcl::PointCloud<pcl::PointXYZI>
// does).  Poses are exposed as plain arrays in the reference's storage order {qx,qy,qz,qw,tx,ty,tz}
// (m_para_buffer_RT, point_cloud_registration.hpp:51-56); wrap them in Eigen::Map<> in the node if desired.
//
// See INTEGRATION.md for the two-line change in laser_feature_extractor.hpp / laser_mapping.hpp.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "loam_livox_hip.h"

namespace loam_livox_hip {

inline void check(int rc, const char *what)
{
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + ll_last_error());
}

template <class Cloud>
inline std::vector<float> cloud_to_xyzi(const Cloud &c)
{
    std::vector<float> v(c.points.size() * 4);
    for (size_t i = 0; i < c.points.size(); i++) {
        v[4 * i + 0] = c.points[i].x;
        v[4 * i + 1] = c.points[i].y;
        v[4 * i + 2] = c.points[i].z;
        v[4 * i + 3] = c.points[i].intensity;
    }
    return v;
}

template <class Cloud>
inline void xyzi_to_cloud(const float *v, int n, Cloud &c)
{
    c.points.resize(n);
    for (int i = 0; i < n; i++) {
        c.points[i].x = v[4 * i + 0];
        c.points[i].y = v[4 * i + 1];
        c.points[i].z = v[4 * i + 2];
        c.points[i].intensity = v[4 * i + 3];
    }
}

// ------------------------------------------------------------------------------------------------------------
class Livox_laser {
   public:
    // public tunables of the reference class (livox_feature_extractor.hpp:143-167); set them before the first
    // extract_laser_features() call, exactly where laser_feature_extractor.hpp:152-154,854,859 sets them
    float max_fov = 17;
    float m_time_internal_pts = 1.0e-5f;
    float thr_corner_curvature = 0.05f;
    float thr_surface_curvature = 0.01f;
    float minimum_view_angle = 10;
    float m_livox_min_allow_dis = 1.0f;
    float m_livox_min_sigma = 7e-3f;
    int m_input_points_size = 0;
    int piecewise_number = 3;  // common/piecewise_number (laser_feature_extractor.hpp:142)
    int device = 0;
    int max_points = 100000;

    struct Pt_infos {  // the fields callers read (livox_feature_extractor.hpp:118-133)
        int pt_type, pt_label, idx;
        float time_stamp, depth_sq2, curvature, view_angle;
    };
    std::vector<Pt_infos> m_pts_info_vec;

    ~Livox_laser()
    {
        if (h_) ll_fe_destroy(h_);
    }

    // std::vector<pcl::PointCloud<PointXYZI>> extract_laser_features(cloud, time_stamp), LFE:722.
    // The caller only uses the number of petal clouds and the first point of / last point of a few of them
    // (laser_feature_extractor.hpp:287-322), so every returned cloud holds exactly those two points.
    template <class Cloud>
    std::vector<Cloud> extract_laser_features(Cloud &laserCloudIn, double time_stamp = -1)
    {
        ensure_handle();
        raw_ = cloud_to_xyzi(laserCloudIn);
        const int n = (int)laserCloudIn.points.size();
        m_input_points_size = n;
        int32_t n_clouds = 0;
        check(ll_fe_extract(h_, raw_.data(), n, time_stamp, &n_clouds), "ll_fe_extract");
        std::vector<int32_t> type(n), label(n), first(max_points / 50 + 8), last(max_points / 50 + 8);
        std::vector<float> depth(n), curv(n), view(n), ts(n);
        check(ll_fe_labels(h_, 0, type.data(), label.data(), depth.data(), nullptr, curv.data(), view.data(), ts.data(), nullptr),
              "ll_fe_labels");
        m_pts_info_vec.resize(n);
        for (int i = 0; i < n; i++) m_pts_info_vec[i] = Pt_infos{type[i], label[i], i, ts[i], depth[i], curv[i], view[i]};
        int32_t ns = 0, cl = 0, npc = 0;
        check(ll_fe_splits(h_, 0, nullptr, &ns, &cl, &npc, first.data(), last.data(), nullptr, nullptr), "ll_fe_splits");
        std::vector<Cloud> out((size_t)npc);
        for (int s = 0; s < npc; s++) {
            out[s].points.resize(first[s] == last[s] ? 1 : 2);
            out[s].points.front() = laserCloudIn.points[first[s]];
            out[s].points.back() = laserCloudIn.points[last[s]];
        }
        return out;
    }

    // Pt_infos *find_pt_info(const T &pt), LFE:206-217: first inserted point with the same xyz
    template <class P>
    Pt_infos *find_pt_info(const P &pt)
    {
        for (size_t i = 0; i < m_pts_info_vec.size(); i++)
            if (raw_[4 * i] == pt.x && raw_[4 * i + 1] == pt.y && raw_[4 * i + 2] == pt.z) return &m_pts_info_vec[i];
        throw std::runtime_error("find_pt_info: point not in the current scan (assert at livox_feature_extractor.hpp:214)");
    }

    // void get_features(pc_corners, pc_surface, pc_full_res, minimum_blur, maximum_blur), LFE:219
    template <class Cloud>
    void get_features(Cloud &pc_corners, Cloud &pc_surface, Cloud &pc_full_res, float minimum_blur = 0.0f, float maximum_blur = 0.3f)
    {
        const int n = m_input_points_size;
        std::vector<int32_t> ci(n), si(n), fi(n);
        std::vector<float> cc((size_t)n * 4), sc((size_t)n * 4);
        int32_t nc = 0, ns = 0, nf = 0;
        check(ll_fe_select(h_, minimum_blur, maximum_blur, ci.data(), &nc, si.data(), &ns, fi.data(), &nf, cc.data(), sc.data()),
              "ll_fe_select");
        xyzi_to_cloud(cc.data(), nc, pc_corners);
        xyzi_to_cloud(sc.data(), ns, pc_surface);
        pc_full_res.points.resize(nf);
        for (int k = 0; k < nf; k++) {  // pc_full keeps every in-window point, intensity := time stamp (LFE:263-265)
            const int i = fi[k];
            pc_full_res.points[k].x = raw_[4 * i];
            pc_full_res.points[k].y = raw_[4 * i + 1];
            pc_full_res.points[k].z = raw_[4 * i + 2];
            pc_full_res.points[k].intensity = m_pts_info_vec[i].time_stamp;
        }
    }

   private:
    void ensure_handle()
    {
        if (h_) return;
        ll_fe_params p;
        ll_fe_default_params(&p);
        p.thr_corner_curvature = thr_corner_curvature;
        p.thr_surface_curvature = thr_surface_curvature;
        p.minimum_view_angle = minimum_view_angle;
        p.livox_min_allow_dis = m_livox_min_allow_dis;
        p.livox_min_sigma = m_livox_min_sigma;
        p.max_fov = max_fov;
        p.time_internal_pts = m_time_internal_pts;
        p.device = device;
        p.max_points = max_points;
        p.max_scans = 1;
        p.piecewise_number = piecewise_number;
        check(ll_fe_create(&p, &h_), "ll_fe_create");
    }
    ll_fe *h_ = nullptr;
    std::vector<float> raw_;
};

// ------------------------------------------------------------------------------------------------------------
// pcl::VoxelGrid<PointType> as the node shells use it: setLeafSize / setInputCloud / filter
// (laser_feature_extractor.hpp:192-193,372-381; laser_mapping.hpp:742-743,1367-1373,1434-1437,533-537).
// CloudPtr is anything that dereferences to a cloud (std::shared_ptr, boost::shared_ptr, raw pointer).
template <class Cloud>
class VoxelGrid {
   public:
    int device = 0;
    int max_points = 0;  // 0: sized for the first cloud seen (grown on demand)
    int status = 0;      // of the last filter(): 0 filtered, 1 leaf too small (copy of the input, like PCL), 2 no finite point

    ~VoxelGrid()
    {
        if (h_) ll_voxel_destroy(h_);
    }
    void setLeafSize(float lx, float ly, float lz)
    {
        leaf_[0] = lx;
        leaf_[1] = ly;
        leaf_[2] = lz;
    }
    template <class CloudPtr>
    void setInputCloud(const CloudPtr &cloud)
    {
        in_ = cloud_to_xyzi(*cloud);
    }
    void filter(Cloud &output)  // `output` may be the input cloud itself (the reference filters in place)
    {
        const int32_t n = (int32_t)(in_.size() / 4);
        if (n == 0) {
            output.points.clear();
            status = 2;
            return;
        }
        if (!h_ || n > cap_) {
            if (h_) ll_voxel_destroy(h_);
            h_ = nullptr;
            cap_ = n > max_points ? n : max_points;
            check(ll_voxel_create(device, 1, cap_, &h_), "ll_voxel_create");
        }
        std::vector<float> out(in_.size());
        int32_t n_out = 0, st = 0;
        check(ll_voxel_filter(h_, 1, in_.data(), &n, n, leaf_, out.data(), &n_out, &st), "ll_voxel_filter");
        status = st;
        xyzi_to_cloud(out.data(), n_out, output);
    }

   private:
    ll_voxel *h_ = nullptr;
    int cap_ = 0;
    float leaf_[3] = {0.4f, 0.4f, 0.4f};
    std::vector<float> in_;
};

// ------------------------------------------------------------------------------------------------------------
// The history match buffer of Laser_mapping (m_matching_mode == 0): m_laser_cloud_corner_history /
// m_laser_cloud_surface_history with the add-frame rule and FIFO of laser_mapping.hpp:1417-1478, and
// update_buff_for_matching's history branch (laser_mapping.hpp:517-546), resident on the device.  refresh() rebuilds the
// search grids of the ll_map the registrar uses, so the two KdTreeFLANN::setInputCloud calls (:544-545) disappear.
class History_buffer {
   public:
    History_buffer(int maximum_history_size, int max_points_per_frame, float line_res, float plane_res, int device = 0)
    {
        check(ll_history_create(device, maximum_history_size, max_points_per_frame, line_res, plane_res, &h_), "ll_history_create");
    }
    ~History_buffer()
    {
        if (h_) ll_history_destroy(h_);
    }
    History_buffer(const History_buffer &) = delete;
    History_buffer &operator=(const History_buffer &) = delete;

    // "Add new frame" (laser_mapping.hpp:1417-1478): features in the sensor frame, pose = {qx,qy,qz,qw,tx,ty,tz} of the
    // accepted registration.  Returns true when the frame was pushed (false: "Reject add history").
    template <class Cloud>
    bool add(const Cloud &corner_stack, const Cloud &surf_stack, const double pose[7], double history_add_t_step = 0.0,
             double history_add_angle_step = 0.0)
    {
        const std::vector<float> c = cloud_to_xyzi(corner_stack), s = cloud_to_xyzi(surf_stack);
        int32_t added = 0;
        check(ll_history_add(h_, c.data(), (int32_t)(c.size() / 4), s.data(), (int32_t)(s.size() / 4), pose, history_add_t_step,
                             history_add_angle_step, &added),
              "ll_history_add");
        return added != 0;
    }
    // update_buff_for_matching(): concatenation of the history -> VoxelGrid -> search grids of `map`
    void refresh(ll_map *map, int64_t *n_corner = nullptr, int64_t *n_surf = nullptr)
    {
        check(ll_history_refresh(h_, map, n_corner, n_surf), "ll_history_refresh");
    }
    // m_laser_cloud_corner_from_map_last / m_laser_cloud_surf_from_map_last of the last refresh (for publishing / saving)
    template <class Cloud>
    void map_cloud(int kind, Cloud &out)
    {
        const int64_t n = ll_history_map_cloud(h_, kind, nullptr, 0);
        std::vector<float> v((size_t)(n > 0 ? n : 0) * 4);
        if (n > 0 && ll_history_map_cloud(h_, kind, v.data(), n) < 0) check(-1, "ll_history_map_cloud");
        xyzi_to_cloud(v.data(), (int)n, out);
    }
    int size() const { return ll_history_size(h_); }

    // m_pt_cell_map_corners / m_pt_cell_map_planes (laser_mapping.hpp:274-275, 617-624) for m_matching_mode == 1: once
    // enabled, every add() also appends the frame to the two cell maps (laser_mapping.hpp:1492-1493)
    void enable_cell_map(int64_t max_points, float m_pt_cell_resolution = 1.0f, int m_para_threshold_cell_revisit = 5000)
    {
        check(ll_history_enable_cell_map(h_, max_points, m_pt_cell_resolution, m_para_threshold_cell_revisit), "ll_history_enable_cell_map");
    }
    // update_buff_for_matching(), cell branch (laser_mapping.hpp:471-546): pose = m_q_w_curr / m_t_w_curr
    void refresh_cells(ll_map *map, const double pose[7], float m_maximum_search_range_corner = 100.0f,
                       float m_maximum_search_range_surface = 100.0f, float m_maximum_in_fov_angle = 30.0f, int m_down_sample_replace = 1,
                       int64_t *n_corner = nullptr, int64_t *n_surf = nullptr)
    {
        check(ll_history_refresh_cells(h_, map, pose, m_maximum_search_range_corner, m_maximum_search_range_surface, m_maximum_in_fov_angle,
                                       m_down_sample_replace, n_corner, n_surf),
              "ll_history_refresh_cells");
    }
    // Points_cloud_map::get_cells_size() and the number of points held, per feature kind (0 corner, 1 surface)
    void cell_map_size(int kind, int64_t *n_cells, int64_t *n_points)
    {
        ll_cellmap *c = ll_history_cell_map(h_, kind);
        if (!c) check(-1, "ll_history_cell_map");
        check(ll_cellmap_stats(c, n_cells, n_points, nullptr), "ll_cellmap_stats");
    }

   private:
    ll_history *h_ = nullptr;
};

// ------------------------------------------------------------------------------------------------------------
// Points_cloud_map<float> (cell_map_keyframe.hpp:477-790) with the cell statistics and key-frame descriptors the loop
// detection reads (determine_feature :436-473, Maps_keyframe::analyze :1385-1493), resident on the device.
class Points_cloud_map {
   public:
    enum Feature_type { e_feature_sphere = 0, e_feature_line = 1, e_feature_plane = 2 };  // cell_map_keyframe.hpp:46-51

    explicit Points_cloud_map(int64_t max_points, float resolution = 1.0f, int m_minimum_revisit_threshold = 2147483647, int device = 0)
    {
        check(ll_cellmap_create(device, max_points, resolution, m_minimum_revisit_threshold, &h_), "ll_cellmap_create");
    }
    ~Points_cloud_map()
    {
        if (h_) ll_cellmap_destroy(h_);
    }
    Points_cloud_map(const Points_cloud_map &) = delete;
    Points_cloud_map &operator=(const Points_cloud_map &) = delete;

    template <class Cloud>
    void append_cloud(const Cloud &cloud)  // :619-672 (intensity is not kept, :82)
    {
        const std::vector<float> v = cloud_to_xyzi(cloud);
        check(ll_cellmap_append(h_, v.data(), (int32_t)(v.size() / 4)), "ll_cellmap_append");
    }
    int64_t get_cells_size() const  // :551-554
    {
        int64_t n = 0;
        check(ll_cellmap_stats(h_, &n, nullptr, nullptr), "ll_cellmap_stats");
        return n;
    }
    // m_feature_type / m_feature_vector of every cell, in ascending cell-index order
    void determine_features(std::vector<int32_t> &feature_type, std::vector<float> &feature_vector)
    {
        const int64_t n = get_cells_size();
        feature_type.assign((size_t)n, 0);
        feature_vector.assign((size_t)n * 3, 0.f);
        check(ll_cellmap_features(h_, feature_type.data(), feature_vector.data(), nullptr, nullptr, nullptr, n), "ll_cellmap_features");
    }
    // m_feature_img_line, m_feature_img_plane, m_feature_img_line_roi, m_feature_img_plane_roi (60 x 60 each, row = phi bin)
    // and m_ratio_nonzero_line / _plane of the four histograms; the map stands for the key frame's cell set
    void analyze(std::vector<float> &images, float ratio_nonzero[4], float roi_ratio = 0.9f)
    {
        images.assign((size_t)4 * 60 * 60, 0.f);
        check(ll_cellmap_keyframe_images(h_, roi_ratio, images.data(), ratio_nonzero, nullptr, nullptr, nullptr), "ll_cellmap_keyframe_images");
    }
    // Maps_keyframe::max_similiarity_of_two_image (:1155-1196) of two 60 x 60 images
    static float max_similiarity_of_two_image(const float *img_a, const float *img_b, int device = 0)
    {
        float s = 0.f;
        check(ll_keyframe_similarity(device, img_a, img_b, &s), "ll_keyframe_similarity");
        return s;
    }
    ll_cellmap *handle() { return h_; }

   private:
    ll_cellmap *h_ = nullptr;
};

// ------------------------------------------------------------------------------------------------------------
class Point_cloud_registration {
   public:
    // configuration fields with the reference names and defaults (point_cloud_registration.hpp:45-103)
    int ICP_PLANE = 1, ICP_LINE = 1;
    int IF_LINE_FEATURE_CHECK = 0, IF_PLANE_FEATURE_CHECK = 0;  // PCR:46,48
    int m_if_motion_deblur = 0;
    int m_current_frame_index = 0;
    int m_mapping_init_accumulate_frames = 100;
    float m_para_max_angular_rate = 200.0f / 50.0f;
    float m_para_max_speed = 100.0f / 50.0f;
    float m_max_final_cost = 100.0f;
    int m_para_icp_max_iterations = 20;
    int m_para_cere_max_iterations = 100;
    int m_para_cere_prerun_times = 2;
    float m_minimum_pt_time_stamp = 0, m_maximum_pt_time_stamp = 1.0f;
    double m_minimum_icp_R_diff = 0.01, m_minimum_icp_T_diff = 0.01;
    double m_inliner_dis = 0.02, m_inlier_ratio = 0.80;
    double m_maximum_dis_plane_for_match = 50.0, m_maximum_dis_line_for_match = 2.0;
    int m_maximum_allow_residual_block = 100000;
    int m_subsample_seed = 1;  // seed of the reproducible stand-in for m_rand_float (PCR:104); 0 = refuse to sub-sample
    // state: {qx,qy,qz,qw,tx,ty,tz}
    double m_para_buffer_RT[7] = {0, 0, 0, 1, 0, 0, 0};        // m_q_w_curr / m_t_w_curr
    double m_para_buffer_RT_last[7] = {0, 0, 0, 1, 0, 0, 0};   // m_q_w_last / m_t_w_last
    double m_para_buffer_incremental[7] = {0, 0, 0, 1, 0, 0, 0};
    double m_inlier_threshold = 0, m_angular_diff = 0, m_t_diff = 0;
    ll_reg_report m_final_opt_summary{};
    int device = 0;
    int max_features = 100000;

    ~Point_cloud_registration()
    {
        if (reg_) ll_reg_destroy(reg_);
        if (map_) ll_map_destroy(map_);
    }

    // int find_out_incremental_transfrom(map_corner, map_surf, kd_corner, kd_surf, scan_corner, scan_surf), PCR:163.
    // The KdTreeFLANN arguments are accepted and ignored: the device grid replaces them.  The map is re-uploaded
    // only when the cloud objects (address + size) change, mirroring the match-buffer refresh of
    // laser_mapping.hpp:460-566.
    template <class CloudPtr, class KdTree>
    int find_out_incremental_transfrom(CloudPtr map_corner, CloudPtr map_surf, KdTree &, KdTree &, CloudPtr scan_corner, CloudPtr scan_surf)
    {
        return find_out_incremental_transfrom(map_corner, map_surf, scan_corner, scan_surf);
    }

    // 4-argument overload, PCR:585-605
    template <class CloudPtr>
    int find_out_incremental_transfrom(CloudPtr map_corner, CloudPtr map_surf, CloudPtr scan_corner, CloudPtr scan_surf)
    {
        upload_if_changed(LL_MAP_CORNER, *map_corner, key_[0]);
        upload_if_changed(LL_MAP_SURF, *map_surf, key_[1]);
        return find_out_incremental_transfrom(scan_corner, scan_surf);
    }

    // The search structure the registrar matches against.  History_buffer::refresh( pc_reg.map() ) /
    // refresh_cells( pc_reg.map(), ... ) rebuild it on the device (update_buff_for_matching, laser_mapping.hpp:460-566);
    // the 2-argument form below then registers against it without a map cloud ever crossing the bus.
    ll_map *map()
    {
        if (!map_) check(ll_map_create(device, &map_), "ll_map_create");
        return map_;
    }

    template <class CloudPtr>
    int find_out_incremental_transfrom(CloudPtr scan_corner, CloudPtr scan_surf)
    {
        if (!reg_) check(ll_reg_create(device, 1, max_features, &reg_), "ll_reg_create");
        map();
        const std::vector<float> c = cloud_to_xyzi(*scan_corner), s = cloud_to_xyzi(*scan_surf);
        ll_reg_params p;
        ll_reg_default_params(&p);
        p.if_motion_deblur = m_if_motion_deblur;
        p.icp_max_iterations = m_para_icp_max_iterations;
        p.ceres_max_iterations = m_para_cere_max_iterations;
        p.ceres_prerun_times = m_para_cere_prerun_times;
        p.icp_line = ICP_LINE;
        p.icp_plane = ICP_PLANE;
        p.if_line_feature_check = IF_LINE_FEATURE_CHECK;
        p.if_plane_feature_check = IF_PLANE_FEATURE_CHECK;
        p.current_frame_index = m_current_frame_index;
        p.mapping_init_accumulate_frames = m_mapping_init_accumulate_frames;
        p.maximum_allow_residual_block = m_maximum_allow_residual_block;
        p.subsample_seed = m_subsample_seed;
        p.maximum_dis_line_for_match = m_maximum_dis_line_for_match;
        p.maximum_dis_plane_for_match = m_maximum_dis_plane_for_match;
        p.inliner_dis = m_inliner_dis;
        p.inlier_ratio = m_inlier_ratio;
        p.minimum_icp_R_diff = m_minimum_icp_R_diff;
        p.minimum_icp_T_diff = m_minimum_icp_T_diff;
        p.para_max_angular_rate = m_para_max_angular_rate;
        p.para_max_speed = m_para_max_speed;
        p.max_final_cost = m_max_final_cost;
        p.minimum_pt_time_stamp = m_minimum_pt_time_stamp;
        p.maximum_pt_time_stamp = m_maximum_pt_time_stamp;
        ll_reg_report rep;
        const int ret = ll_reg_solve(reg_, map_, c.data(), (int)(c.size() / 4), s.data(), (int)(s.size() / 4), &p, m_para_buffer_RT_last,
                                     m_para_buffer_RT, m_para_buffer_incremental, &rep);
        check(ret, "ll_reg_solve");
        m_inlier_threshold = rep.inlier_threshold;
        m_angular_diff = rep.angular_diff_deg;
        m_t_diff = rep.t_diff;
        if (ret == 1) m_final_opt_summary = rep;  // PCR:574
        return ret;
    }

    // unsigned int pointcloudAssociateToMap(pc_in, pt_out, if_undistore = 0), PCR:673-685
    template <class Cloud>
    unsigned int pointcloudAssociateToMap(const Cloud &pc_in, Cloud &pt_out, int /*if_undistore*/ = 0)
    {
        if (!reg_) check(ll_reg_create(device, 1, max_features, &reg_), "ll_reg_create");
        const std::vector<float> in = cloud_to_xyzi(pc_in);
        std::vector<float> out(in.size());
        check(ll_cloud_transform(reg_, in.data(), out.data(), (int)(in.size() / 4), m_para_buffer_RT), "ll_cloud_transform");
        xyzi_to_cloud(out.data(), (int)(in.size() / 4), pt_out);
        return (unsigned int)(in.size() / 4);
    }

   private:
    struct Key {
        const void *p = nullptr;
        size_t n = 0;
    };
    template <class Cloud>
    void upload_if_changed(int kind, const Cloud &c, Key &k)
    {
        if (k.p == (const void *)c.points.data() && k.n == c.points.size()) return;
        map();
        const std::vector<float> v = cloud_to_xyzi(c);
        check(ll_map_upload(map_, kind, v.data(), 4, (int64_t)c.points.size(), 0.0f), "ll_map_upload");
        k.p = (const void *)c.points.data();
        k.n = c.points.size();
    }
    ll_reg *reg_ = nullptr;
    ll_map *map_ = nullptr;
    Key key_[2];
};

}  // namespace loam_livox_hip

// loam_livox_adapter.hpp -- header-only C++ adapter that gives the reference's node shells the call surface they
// already use (hku-mars/loam_livox), backed by the C ABI of libloamlivox_hip.so:
//
//   loam_livox_hip::Livox_laser               <->  class Livox_laser               source/livox_feature_extractor.hpp:77
//   loam_livox_hip::Point_cloud_registration  <->  class Point_cloud_registration  source/point_cloud_registration.hpp:38
//
// The adapter is templated on the cloud type so that it compiles with or without PCL: any type with a
// `points` std::vector whose elements have float members x, y, z, intensity works (pcl::PointCloud<pcl::PointXYZI>
// does).  Poses are the reference's own members (m_q_w_curr, m_t_w_curr, m_q_w_last, m_t_w_last, m_q_w_incre, m_t_w_incre;
// Eigen objects when Eigen is on the include path) next to the raw buffers in the reference's storage order
// {qx,qy,qz,qw,tx,ty,tz} (m_para_buffer_RT, point_cloud_registration.hpp:51-56).
//
// See INTEGRATION.md for the two-line change in laser_feature_extractor.hpp / laser_mapping.hpp.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "loam_livox_hip.h"

// The registrar's pose members (m_q_w_curr, m_t_w_curr, m_q_w_last, m_t_w_last, m_q_w_incre, m_t_w_incre,
// point_cloud_registration.hpp:53-56,75-76) are Eigen objects in the reference and the node assigns Eigen objects to them
// (laser_mapping.hpp:1290-1294, 1498-1499).  With Eigen on the include path they ARE Eigen objects here; without it
// (no-dependency builds, the C++ smoke test) they are the small look-alikes below.
#if defined(__has_include)
#if __has_include(<Eigen/Geometry>) && !defined(LOAM_LIVOX_ADAPTER_NO_EIGEN)
#include <Eigen/Core>
#include <Eigen/Geometry>
#define LOAM_LIVOX_ADAPTER_EIGEN 1
#endif
#endif

namespace loam_livox_hip {

#if defined(LOAM_LIVOX_ADAPTER_EIGEN)
typedef Eigen::Quaterniond Quaterniond;
typedef Eigen::Vector3d Vector3d;
typedef Eigen::Map<Eigen::Quaterniond> Quaterniond_map;
typedef Eigen::Map<Eigen::Vector3d> Vector3d_map;
#else
struct Quaterniond {  // storage x, y, z, w like Eigen
    double c[4] = {0, 0, 0, 1};
    double &x() { return c[0]; }
    double &y() { return c[1]; }
    double &z() { return c[2]; }
    double &w() { return c[3]; }
    double x() const { return c[0]; }
    double y() const { return c[1]; }
    double z() const { return c[2]; }
    double w() const { return c[3]; }
    void setIdentity() { c[0] = c[1] = c[2] = 0, c[3] = 1; }
};
struct Vector3d {
    double c[3] = {0, 0, 0};
    double &operator()(int i) { return c[i]; }
    double operator()(int i) const { return c[i]; }
    double &x() { return c[0]; }
    double &y() { return c[1]; }
    double &z() { return c[2]; }
    double x() const { return c[0]; }
    double y() const { return c[1]; }
    double z() const { return c[2]; }
    void setZero() { c[0] = c[1] = c[2] = 0; }
};
struct Quaterniond_map {
    double *p;
    explicit Quaterniond_map(double *q) : p(q) {}
    double &x() { return p[0]; }
    double &y() { return p[1]; }
    double &z() { return p[2]; }
    double &w() { return p[3]; }
    Quaterniond_map &operator=(const Quaterniond &q)
    {
        for (int i = 0; i < 4; i++) p[i] = q.c[i];
        return *this;
    }
    operator Quaterniond() const
    {
        Quaterniond q;
        for (int i = 0; i < 4; i++) q.c[i] = p[i];
        return q;
    }
};
struct Vector3d_map {
    double *p;
    explicit Vector3d_map(double *q) : p(q) {}
    double &operator()(int i) { return p[i]; }
    Vector3d_map &operator=(const Vector3d &v)
    {
        for (int i = 0; i < 3; i++) p[i] = v.c[i];
        return *this;
    }
    operator Vector3d() const
    {
        Vector3d v;
        for (int i = 0; i < 3; i++) v.c[i] = p[i];
        return v;
    }
};
#endif

// Pointer members the node sets but the device path has no use for (m_logger_common, m_logger_pcd, m_logger_timer,
// m_timer: point_cloud_registration.hpp:78-82, set at laser_mapping.hpp:1271-1274, scene_alignment.hpp:234-236): any
// pointer can be assigned, it is kept as-is.
struct Any_ptr {
    const void *p = nullptr;
    template <class T>
    Any_ptr &operator=(T *q)
    {
        p = (const void *)q;
        return *this;
    }
};
// m_kdtree_corner_from_map / m_kdtree_surf_from_map (point_cloud_registration.hpp:72-73): the device grid replaces the
// k-d trees, assignments are accepted and dropped.
struct Any_sink {
    template <class T>
    Any_sink &operator=(const T &)
    {
        return *this;
    }
};

inline void check(int rc, const char *what)
{
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + ll_last_error());
}

// Runtime hints of the process, applied once before the adapter's first *_create (= before its first HIP call): 16 hardware queues
// instead of the runtime's default of 4, so that registrations in flight on different handles do not share a queue
// (ll_runtime_hint_hw_queues in loam_livox_hip.h).  The caller's environment always wins; a host that
// initialises HIP before constructing any adapter object must export GPU_MAX_HW_QUEUES itself.  -DLOAM_LIVOX_HIP_NO_RUNTIME_HINTS
// leaves the process environment alone.
inline void runtime_hints()
{
#ifndef LOAM_LIVOX_HIP_NO_RUNTIME_HINTS
    static std::once_flag once;
    std::call_once(once, [] { (void)ll_runtime_hint_hw_queues(16); });
#endif
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

    // std::vector<pcl::PointCloud<PointXYZI>> extract_laser_features(cloud, time_stamp), LFE:722: the petal clouds of
    // split_laser_scan (LFE:657-719) -- a new cloud whenever the petal angle changes, the last petal dropped (:681), points
    // masked 000 / too-near / nan removed, intensity = idx / N (set_intensity(e_I_motion_blur), :283-286), empty petals
    // removed -- rebuilt on the host from the device's per-point planes.
    template <class Cloud>
    std::vector<Cloud> extract_laser_features(Cloud &laserCloudIn, double time_stamp = -1)
    {
        ensure_handle();
        raw_ = cloud_to_xyzi(laserCloudIn);
        const int n = (int)laserCloudIn.points.size();
        m_input_points_size = n;
        first_of_.clear();
        int32_t n_clouds = 0;
        check(ll_fe_extract(h_, raw_.data(), n, time_stamp, &n_clouds), "ll_fe_extract");
        std::vector<int32_t> type(n), label(n);
        std::vector<float> depth(n), curv(n), view(n), ts(n), pang(n);
        check(ll_fe_labels(h_, 0, type.data(), label.data(), depth.data(), nullptr, curv.data(), view.data(), ts.data(), pang.data()),
              "ll_fe_labels");
        m_pts_info_vec.resize(n);
        for (int i = 0; i < n; i++) m_pts_info_vec[i] = Pt_infos{type[i], label[i], i, ts[i], depth[i], curv[i], view[i]};
        int32_t ns = 0, cl = 0, npc = 0;
        check(ll_fe_splits(h_, 0, nullptr, &ns, &cl, &npc, nullptr, nullptr, nullptr, nullptr), "ll_fe_splits");
        std::vector<Cloud> out;
        if (cl == 0) return out;  // fewer than 6 split entries (LFE:572, 759-762)
        int scan_idx = 0;
        std::vector<Cloud> petals((size_t)cl);
        for (int i = 0; i < n; i++) {
            if (i > 0 && pang[i] != pang[i - 1]) scan_idx++;
            if (scan_idx >= cl) break;  // cannot happen: clutter_size counts the angle changes + 1
            if (type[i] & (1 | 2 | 32)) continue;  // e_pt_000 | e_pt_too_near | e_pt_nan, LFE:684-688
            auto pt = laserCloudIn.points[i];
            // set_intensity(e_I_motion_blur) reads find_pt_info(pt)->idx (LFE:278-286, 706): the index of the FIRST inserted point
            // with these coordinates (the unordered_map keeps the first of equal keys, LFE:478)
            pt.intensity = (float)find_pt_info(pt)->idx / (float)n;
            petals[scan_idx].points.push_back(pt);
        }
        for (int s = 0; s < scan_idx && s < cl; s++)  // resize(scan_idx): the last petal is dropped, LFE:681
            if (!petals[s].points.empty()) out.push_back(petals[s]);
        if ((int)out.size() != npc) throw std::runtime_error("extract_laser_features: petal count differs from the device's");
        return out;
    }

    // Pt_infos *find_pt_info(const T &pt), LFE:206-217: the first inserted point with the same xyz (the unordered_map keeps
    // the first of equal keys, LFE:478).  A hash index over the scan is built at the first look-up after an extraction.
    template <class P>
    Pt_infos *find_pt_info(const P &pt)
    {
        if (first_of_.empty() && !m_pts_info_vec.empty()) {
            size_t cap = 16;
            while (cap < 2 * m_pts_info_vec.size()) cap <<= 1;
            first_of_.assign(cap, -1);
            for (size_t i = 0; i < m_pts_info_vec.size(); i++) {
                size_t h = hash_xyz(raw_[4 * i], raw_[4 * i + 1], raw_[4 * i + 2]) & (cap - 1);
                for (;; h = (h + 1) & (cap - 1)) {
                    const int j = first_of_[h];
                    if (j < 0) {
                        first_of_[h] = (int)i;
                        break;
                    }
                    if (raw_[4 * j] == raw_[4 * i] && raw_[4 * j + 1] == raw_[4 * i + 1] && raw_[4 * j + 2] == raw_[4 * i + 2]) break;
                }
            }
        }
        if (!first_of_.empty()) {
            const size_t cap = first_of_.size();
            for (size_t h = hash_xyz(pt.x, pt.y, pt.z) & (cap - 1);; h = (h + 1) & (cap - 1)) {
                const int j = first_of_[h];
                if (j < 0) break;
                if (raw_[4 * j] == pt.x && raw_[4 * j + 1] == pt.y && raw_[4 * j + 2] == pt.z) return &m_pts_info_vec[j];
            }
        }
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
        runtime_hints();
        check(ll_fe_create(&p, &h_), "ll_fe_create");
    }
    static size_t hash_xyz(float x, float y, float z)
    {
        uint32_t a, b, c;
        x = x + 0.0f;  // -0.0f == 0.0f must hash alike (the reference compares with ==, pcl_tools.hpp:32-36)
        y = y + 0.0f;
        z = z + 0.0f;
        std::memcpy(&a, &x, 4);
        std::memcpy(&b, &y, 4);
        std::memcpy(&c, &z, 4);
        uint64_t h = (uint64_t)a * 0x9E3779B97F4A7C15ull ^ ((uint64_t)b << 21) * 0xC2B2AE3D27D4EB4Full ^ (uint64_t)c * 0x165667B19E3779F9ull;
        return (size_t)(h ^ (h >> 29));
    }
    ll_fe *h_ = nullptr;
    std::vector<float> raw_;
    std::vector<int> first_of_;
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

    VoxelGrid() {}
    // The node copies its filters by value into every call (laser_mapping.hpp:465-466, 1325-1326): a copy takes the configuration
    // and creates its own device handle on first use.
    VoxelGrid(const VoxelGrid &o) : device(o.device), max_points(o.max_points), status(0)
    {
        leaf_[0] = o.leaf_[0];
        leaf_[1] = o.leaf_[1];
        leaf_[2] = o.leaf_[2];
    }
    VoxelGrid &operator=(const VoxelGrid &o)
    {
        if (this != &o) {
            device = o.device;
            max_points = o.max_points;
            leaf_[0] = o.leaf_[0];
            leaf_[1] = o.leaf_[1];
            leaf_[2] = o.leaf_[2];
            in_.clear();
        }
        return *this;
    }
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
            runtime_hints();
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
        runtime_hints();
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
    // The add-frame rule (laser_mapping.hpp:1439-1451) reads the node's m_q_w_curr / m_t_w_curr, which is still the pose
    // BEFORE the registration whose result add() receives: hand it over first (applies to the next add only).
    void set_gate_pose(const double pose_before_registration[7]) { check(ll_history_set_gate_pose(h_, pose_before_registration), "ll_history_set_gate_pose"); }
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
    // m_matching_mode == 0 never reads the cell maps between frames (laser_mapping.hpp:1492-1493 only appends): feed them beside the mapping
    // loop, on a service thread of the handle, as the reference runs its own map services on threads (:568-594).  Every reader
    // (refresh_cells, cell_map_size, cell_map) waits for the frames handed over so far; sync_cell_maps() does only that.
    void set_cell_map_async(bool enable = true) { check(ll_history_set_cell_map_async(h_, enable ? 1 : 0), "ll_history_set_cell_map_async"); }
    void sync_cell_maps() { check(ll_history_sync_cell_maps(h_), "ll_history_sync_cell_maps"); }
    // the cell map of one feature kind (borrowed: valid as long as this object), e.g. for ll_cellmap_device_view -- the input of a multi-GPU
    // gather of cell maps
    ll_cellmap *cell_map(int kind)
    {
        ll_cellmap *c = ll_history_cell_map(h_, kind);
        if (!c) check(-1, "ll_history_cell_map");
        return c;
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
        : capacity_(max_points)
    {
        runtime_hints();
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
    // append_cloud( pts, &cell_vec ) (:619-672, what the mapping node calls when loop closure is on, laser_mapping.hpp:1442): the
    // reference's cell_vec is a set of cell POINTERS; the cells live on the device here, so the set holds their integer cell
    // indices {ix, iy, iz} (centre = index * box + box / 2, :559-568) -- every cell of the first cloud, afterwards the cells that
    // received at least three points of this cloud (:640-662).  Set arithmetic of Maps_keyframe::add_cells (:1243-1261) works on
    // these as it does on the pointers.
    typedef std::array<int32_t, 3> Cell_index;
    template <class Cloud>
    void append_cloud(const Cloud &cloud, std::set<Cell_index> *cell_vec)
    {
        if (!cell_vec) return append_cloud(cloud);
        const std::vector<float> v = cloud_to_xyzi(cloud);
        const int32_t n = (int32_t)(v.size() / 4);
        std::vector<int32_t> ijk((size_t)(n > 0 ? n : 1) * 3);  // a cloud of n points touches at most n cells
        int64_t n_touched = 0;
        check(ll_cellmap_append_touched(h_, v.data(), n, 3, ijk.data(), (int64_t)(ijk.size() / 3), &n_touched), "ll_cellmap_append_touched");
        cell_vec->clear();
        const int64_t n_listed = n_touched < (int64_t)(ijk.size() / 3) ? n_touched : (int64_t)(ijk.size() / 3);  // (a short buffer truncates; this one never is)
        for (int64_t i = 0; i < n_listed; i++) cell_vec->insert(Cell_index{ijk[3 * i], ijk[3 * i + 1], ijk[3 * i + 2]});
    }
    // The reference's cells live on the heap and the map grows without bound (:619-672); the device map has a capacity: raise it, content,
    // revisit stamps and frame counter kept (no-op when not larger).
    void reserve(int64_t max_points) { check(ll_cellmap_reserve(h_, max_points), "ll_cellmap_reserve"); }
    // the stored points ({x, y, z, 0}, ordered by (cell, insertion)) and the 64-bit cell key of every point where they lie on the device;
    // valid until the next call that changes the map
    void device_view(const float **dev_xyz0, const uint64_t **dev_point_keys, int64_t *n_points, int64_t *n_cells = nullptr)
    {
        check(ll_cellmap_device_view(h_, dev_xyz0, dev_point_keys, n_points, n_cells), "ll_cellmap_device_view");
    }
    // grows the capacity (doubling) so that n_more points fit: the reference's map grows on the heap without bound
    void reserve_for(size_t n_more)
    {
        int64_t n_pts = 0, cap = capacity_;
        check(ll_cellmap_stats(h_, nullptr, &n_pts, nullptr), "ll_cellmap_stats");
        if (cap <= 0) cap = 1;
        while (cap < n_pts + (int64_t)n_more) cap *= 2;
        if (cap != capacity_) {
            reserve(cap);
            capacity_ = cap;
        }
    }
    // find_cells_in_radius( centre, radius ) (:761-788) followed by what service_pub_surround_pts does with the cells
    // (laser_mapping.hpp:1172-1187): every cell's cloud through pcl::VoxelGrid( leaf ) on its own, concatenated in ascending cell order;
    // nothing is stored back.  pose[4..6] is the centre (no field-of-view test: maximum_in_fov_angle >= 360 switches it off).
    template <class Cloud>
    void find_cells_in_radius_filtered(const double pose[7], float radius, float leaf, Cloud &out)
    {
        int64_t n_sel = 0, n_out = 0;
        check(ll_cellmap_query_filter(h_, pose, radius, 360.0f, leaf, 0, &n_sel, &n_out), "ll_cellmap_query_filter");
        std::vector<float> v((size_t)(n_out > 0 ? n_out : 0) * 4);
        if (n_out > 0 && ll_cellmap_result(h_, v.data(), n_out) < 0) check(-1, "ll_cellmap_result");
        xyzi_to_cloud(v.data(), (int)n_out, out);
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
    int64_t capacity_ = 0;
};

// ------------------------------------------------------------------------------------------------------------
// Device handles are expensive to create (hipMalloc of the per-scan scratch) and the node constructs a fresh
// Point_cloud_registration for every scan (laser_mapping.hpp:1348), up to maximum_parallel_thread of them at once
// (:1737-1742).  Objects therefore borrow their ll_reg from a process-wide pool and give it back in the destructor; the
// match-buffer ll_map is shared by all of them (an immutable snapshot is pinned by each solve, see ll_map_upload).
class Handle_pool {
   public:
    static Handle_pool &instance()
    {
        static Handle_pool p;
        return p;
    }
    ll_reg *acquire(int device, int max_features)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].device == device && free_[i].max_features >= max_features) {
                    ll_reg *r = free_[i].reg;
                    free_.erase(free_.begin() + (long)i);
                    return r;
                }
        }
        ll_reg *r = nullptr;
        runtime_hints();
        check(ll_reg_create(device, 1, max_features, &r), "ll_reg_create");
        std::lock_guard<std::mutex> lk(mu_);
        feat_.push_back(std::make_pair(r, Entry{r, device, max_features}));
        return r;
    }
    void release(ll_reg *r)
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (auto &kv : feat_)
            if (kv.first == r) {
                free_.push_back(kv.second);
                return;
            }
    }
    // What the shared match-buffer map of a device holds, as far as this process uploaded it: size and a hash of EVERY point of
    // the cloud (the node allocates new clouds for each scan and deep-copies the match buffer into them, laser_mapping.hpp:1391-1392,
    // 1396-1401: an address says nothing, and an allocator may hand the same address out again for other contents), plus the map's
    // generation right after that upload -- anything else that publishes a structure (History_buffer::refresh) voids the key.
    struct Key {
        size_t n = (size_t)-1;
        uint64_t hash = 0;
        int64_t generation = -1;
    };
    struct Shared_map {
        int device = 0;
        ll_map *map = nullptr;
        Key key[2];
        std::mutex upload_mu;  // upload + pin of one registration are atomic against another thread's upload
    };
    Shared_map &shared(int device)
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (auto &m : maps_)
            if (m->device == device) return *m;
        ll_map *m = nullptr;
        runtime_hints();
        check(ll_map_create(device, &m), "ll_map_create");
        maps_.push_back(std::unique_ptr<Shared_map>(new Shared_map()));
        maps_.back()->device = device;
        maps_.back()->map = m;
        return *maps_.back();
    }
    ll_map *shared_map(int device) { return shared(device).map; }

   private:
    struct Entry {
        ll_reg *reg;
        int device, max_features;
    };
    Handle_pool() {}
    // No HIP calls at static destruction: the runtime may already be gone by then (the handles die with the process).
    ~Handle_pool() {}
    std::mutex mu_;
    std::vector<Entry> free_;
    std::vector<std::pair<ll_reg *, Entry>> feat_;
    std::vector<std::unique_ptr<Shared_map>> maps_;
};

// ------------------------------------------------------------------------------------------------------------
class Point_cloud_registration {
   public:
    // configuration fields with the reference names and defaults (point_cloud_registration.hpp:45-103)
    int ICP_PLANE = 1, ICP_LINE = 1;
    int IF_LINE_FEATURE_CHECK = 0, IF_PLANE_FEATURE_CHECK = 0;  // PCR:46,48
    int line_search_num = 5, plane_search_num = 5;              // PCR:45,47 (the device search is fixed at k = 5)
    int m_if_motion_deblur = 0;
    int m_current_frame_index = 0;
    int m_mapping_init_accumulate_frames = 100;
    float m_last_time_stamp = 0;
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
    int m_if_verbose_screen_printf = 1;  // ADD_SCREEN_PRINTF_OUT_METHOD (PCR:119); nothing is printed here
    // what the node points at its loggers / timer (PCR:78-82, LM:1271-1274, SA:234-236) and the k-d tree members (PCR:72-73)
    Any_ptr m_logger_common, m_logger_pcd, m_logger_timer, m_timer;
    Any_sink m_kdtree_corner_from_map, m_kdtree_surf_from_map;
    // state, with the reference's member names and types (PCR:51-56, 75-76)
    double m_para_buffer_RT[7] = {0, 0, 0, 1, 0, 0, 0};
    double m_para_buffer_RT_last[7] = {0, 0, 0, 1, 0, 0, 0};
    double m_para_buffer_incremental[7] = {0, 0, 0, 1, 0, 0, 0};
    Quaterniond_map m_q_w_incre = Quaterniond_map(m_para_buffer_incremental);
    Vector3d_map m_t_w_incre = Vector3d_map(m_para_buffer_incremental + 4);
    Quaterniond m_q_w_curr, m_q_w_last;
    Vector3d m_t_w_curr, m_t_w_last;
    double m_inlier_threshold = 0, m_angular_diff = 0, m_t_diff = 0;
    // ceres::Solver::Summary look-alike: the fields and reports the node reads (LM:1041, 1511-1512; PCR:555-559)
    struct Opt_summary : ll_reg_report {
        int num_residual_blocks = 0;
        Opt_summary() : ll_reg_report() {}
        std::string BriefReport() const
        {
            char b[256];
            std::snprintf(b, sizeof(b), "loam_livox_hip: ICP iterations %d, LM iterations %d, initial_cost %.6e, final_cost %.6e, blocks %d",
                          icp_iterations, lm_iterations_total, initial_cost, final_cost, n_blocks_last);
            return b;
        }
        std::string FullReport() const { return BriefReport(); }
    };
    Opt_summary summary, m_final_opt_summary;
    int device = 0;
    int max_features = 100000;

    Point_cloud_registration()
    {
        m_q_w_last.setIdentity();
        m_t_w_last.setZero();
        m_q_w_curr.setIdentity();
        m_t_w_curr.setZero();
    }
    // the maps refer to this object's own buffer: copies re-seat them
    Point_cloud_registration(const Point_cloud_registration &o) { *this = o; }
    Point_cloud_registration &operator=(const Point_cloud_registration &o)
    {
        if (this == &o) return *this;
        ICP_PLANE = o.ICP_PLANE, ICP_LINE = o.ICP_LINE;
        IF_LINE_FEATURE_CHECK = o.IF_LINE_FEATURE_CHECK, IF_PLANE_FEATURE_CHECK = o.IF_PLANE_FEATURE_CHECK;
        m_if_motion_deblur = o.m_if_motion_deblur, m_current_frame_index = o.m_current_frame_index;
        m_mapping_init_accumulate_frames = o.m_mapping_init_accumulate_frames, m_last_time_stamp = o.m_last_time_stamp;
        m_para_max_angular_rate = o.m_para_max_angular_rate, m_para_max_speed = o.m_para_max_speed, m_max_final_cost = o.m_max_final_cost;
        m_para_icp_max_iterations = o.m_para_icp_max_iterations, m_para_cere_max_iterations = o.m_para_cere_max_iterations;
        m_para_cere_prerun_times = o.m_para_cere_prerun_times;
        m_minimum_pt_time_stamp = o.m_minimum_pt_time_stamp, m_maximum_pt_time_stamp = o.m_maximum_pt_time_stamp;
        m_minimum_icp_R_diff = o.m_minimum_icp_R_diff, m_minimum_icp_T_diff = o.m_minimum_icp_T_diff;
        m_inliner_dis = o.m_inliner_dis, m_inlier_ratio = o.m_inlier_ratio;
        m_maximum_dis_plane_for_match = o.m_maximum_dis_plane_for_match, m_maximum_dis_line_for_match = o.m_maximum_dis_line_for_match;
        m_maximum_allow_residual_block = o.m_maximum_allow_residual_block, m_subsample_seed = o.m_subsample_seed;
        m_if_verbose_screen_printf = o.m_if_verbose_screen_printf;
        m_logger_common = o.m_logger_common, m_logger_pcd = o.m_logger_pcd, m_logger_timer = o.m_logger_timer, m_timer = o.m_timer;
        for (int i = 0; i < 7; i++) {
            m_para_buffer_RT[i] = o.m_para_buffer_RT[i];
            m_para_buffer_RT_last[i] = o.m_para_buffer_RT_last[i];
            m_para_buffer_incremental[i] = o.m_para_buffer_incremental[i];
        }
        m_q_w_curr = o.m_q_w_curr, m_q_w_last = o.m_q_w_last, m_t_w_curr = o.m_t_w_curr, m_t_w_last = o.m_t_w_last;
        m_inlier_threshold = o.m_inlier_threshold, m_angular_diff = o.m_angular_diff, m_t_diff = o.m_t_diff;
        summary = o.summary, m_final_opt_summary = o.m_final_opt_summary;
        device = o.device, max_features = o.max_features;
        return *this;
    }

    ~Point_cloud_registration()
    {
        if (reg_) Handle_pool::instance().release(reg_);
    }

    // float refine_blur(in_blur, min_blur, max_blur), PCR:128-141
    float refine_blur(float in_blur, const float &min_blur, const float &max_blur) const
    {
        float res = 1.0f;
        if (m_if_motion_deblur) {
            res = (in_blur - min_blur) / (max_blur - min_blur);
            if (!std::isfinite(res) || res > 1.0f) return 1.0f;
        }
        return res;
    }

    // int find_out_incremental_transfrom(map_corner, map_surf, kd_corner, kd_surf, scan_corner, scan_surf), PCR:163.
    // The KdTreeFLANN arguments are accepted and ignored: the device grid replaces them.  The map is re-uploaded only
    // when a cloud changes (address, size or a sample of its contents), mirroring the match-buffer refresh of
    // laser_mapping.hpp:460-566; a solve that is already running keeps the snapshot it started with.
    template <class CloudPtr, class KdTree>
    int find_out_incremental_transfrom(CloudPtr map_corner, CloudPtr map_surf, KdTree &, KdTree &, CloudPtr scan_corner, CloudPtr scan_surf)
    {
        return find_out_incremental_transfrom(map_corner, map_surf, scan_corner, scan_surf);
    }

    // 4-argument overload, PCR:585-605 (returns 1 without registering when a map cloud is empty, :592-601)
    template <class CloudPtr>
    int find_out_incremental_transfrom(CloudPtr map_corner, CloudPtr map_surf, CloudPtr scan_corner, CloudPtr scan_surf)
    {
        // the solve pins the snapshots it is enqueued against: with upload and enqueue under one lock, a thread registers against
        // the clouds it was given even when another thread (laser_mapping.hpp:1737-1742) uploads other ones meanwhile
        Handle_pool::Shared_map &sm = Handle_pool::instance().shared(device);
        std::unique_lock<std::mutex> lk(sm.upload_mu);
        upload_if_changed(sm, LL_MAP_CORNER, *map_corner);
        upload_if_changed(sm, LL_MAP_SURF, *map_surf);
        return register_scan(scan_corner, scan_surf, &lk);
    }

    // The search structure the registrar matches against.  History_buffer::refresh( pc_reg.map() ) /
    // refresh_cells( pc_reg.map(), ... ) rebuild it on the device (update_buff_for_matching, laser_mapping.hpp:460-566);
    // the 2-argument form below then registers against it without a map cloud ever crossing the bus.
    ll_map *map() { return Handle_pool::instance().shared_map(device); }

    template <class CloudPtr>
    int find_out_incremental_transfrom(CloudPtr scan_corner, CloudPtr scan_surf)
    {
        return register_scan(scan_corner, scan_surf, nullptr);
    }

    // void pointAssociateToMap ... see below
   private:
    template <class CloudPtr>
    int register_scan(CloudPtr scan_corner, CloudPtr scan_surf, std::unique_lock<std::mutex> *held_until_enqueued)
    {
        if (!reg_) reg_ = Handle_pool::instance().acquire(device, max_features);
        ll_map *m = map();
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
        pose_from_members();
        ll_reg_report rep;
        // ll_reg_solve in its three steps (upload the scan's features, enqueue = pin the map snapshots + launch, collect)
        const int32_t nc = (int32_t)(c.size() / 4), ns = (int32_t)(s.size() / 4);
        const float dummy[4] = {0, 0, 0, 0};
        check(ll_reg_upload_features(reg_, 1, nc ? c.data() : dummy, &nc, nc > 0 ? nc : 1, ns ? s.data() : dummy, &ns, ns > 0 ? ns : 1),
              "ll_reg_upload_features");
        check(ll_reg_enqueue_uploaded(reg_, m, 1, &p, m_para_buffer_RT_last, m_para_buffer_RT, m_para_buffer_incremental), "ll_reg_enqueue_uploaded");
        if (held_until_enqueued) held_until_enqueued->unlock();
        int32_t ret = 0;
        check(ll_reg_collect(reg_, 1, m_para_buffer_RT, m_para_buffer_incremental, &rep, &ret), "ll_reg_collect");
        members_from_pose();
        m_inlier_threshold = rep.inlier_threshold;
        m_angular_diff = rep.angular_diff_deg;
        m_t_diff = rep.t_diff;
        static_cast<ll_reg_report &>(summary) = rep;
        summary.num_residual_blocks = rep.n_blocks_last;
        if (ret == 1 && !rep.gated) m_final_opt_summary = summary;  // PCR:574
        if (ret == 0) m_last_time_stamp = m_minimum_pt_time_stamp;   // PCR:569
        return ret;
    }

   public:

    // void pointAssociateToMap(pi, po, interpolate_s = 1.0, if_undistore = 0), PCR:622-661.  The node calls it per point
    // with g_if_undistore == 0 (laser_mapping.hpp:80, 1424, 1430): p_w = q_w_curr * p + t_w_curr in double, stored to
    // float, with Eigen's rotation formula  v + w (2 q x v) + q x (2 q x v).  The Rodrigues-interpolated form needs the
    // increment of the registration in progress and lives on the device only (if_undistore != 0 throws).
    template <class P>
    void pointAssociateToMap(P const *const pi, P *const po, double interpolate_s = 1.0, int if_undistore = 0)
    {
        if (!(m_if_motion_deblur == 0 || if_undistore == 0 || interpolate_s == 1.0))
            throw std::runtime_error("pointAssociateToMap: the motion-deblur interpolation (PCR:633-654) is only available inside the registrar");
        const double qx = m_q_w_curr.x(), qy = m_q_w_curr.y(), qz = m_q_w_curr.z(), qw = m_q_w_curr.w();
        const double v[3] = {(double)pi->x, (double)pi->y, (double)pi->z};
        double uv[3] = {qy * v[2] - qz * v[1], qz * v[0] - qx * v[2], qx * v[1] - qy * v[0]};
        uv[0] += uv[0];
        uv[1] += uv[1];
        uv[2] += uv[2];
        const double c[3] = {qy * uv[2] - qz * uv[1], qz * uv[0] - qx * uv[2], qx * uv[1] - qy * uv[0]};
        po->x = (float)((v[0] + qw * uv[0] + c[0]) + m_t_w_curr(0));
        po->y = (float)((v[1] + qw * uv[1] + c[1]) + m_t_w_curr(1));
        po->z = (float)((v[2] + qw * uv[2] + c[2]) + m_t_w_curr(2));
        po->intensity = pi->intensity;
    }

    // unsigned int pointcloudAssociateToMap(pc_in, pt_out, if_undistore = 0), PCR:673-685
    template <class Cloud>
    unsigned int pointcloudAssociateToMap(const Cloud &pc_in, Cloud &pt_out, int /*if_undistore*/ = 0)
    {
        if (!reg_) reg_ = Handle_pool::instance().acquire(device, max_features);
        pose_from_members();
        const std::vector<float> in = cloud_to_xyzi(pc_in);
        std::vector<float> out(in.size());
        check(ll_cloud_transform(reg_, in.data(), out.data(), (int)(in.size() / 4), m_para_buffer_RT), "ll_cloud_transform");
        xyzi_to_cloud(out.data(), (int)(in.size() / 4), pt_out);
        return (unsigned int)(in.size() / 4);
    }

   private:
    void pose_from_members()
    {
        m_para_buffer_RT[0] = m_q_w_curr.x(), m_para_buffer_RT[1] = m_q_w_curr.y(), m_para_buffer_RT[2] = m_q_w_curr.z(), m_para_buffer_RT[3] = m_q_w_curr.w();
        m_para_buffer_RT_last[0] = m_q_w_last.x(), m_para_buffer_RT_last[1] = m_q_w_last.y(), m_para_buffer_RT_last[2] = m_q_w_last.z(),
        m_para_buffer_RT_last[3] = m_q_w_last.w();
        for (int i = 0; i < 3; i++) {
            m_para_buffer_RT[4 + i] = m_t_w_curr(i);
            m_para_buffer_RT_last[4 + i] = m_t_w_last(i);
        }
    }
    void members_from_pose()
    {
        m_q_w_curr.x() = m_para_buffer_RT[0], m_q_w_curr.y() = m_para_buffer_RT[1], m_q_w_curr.z() = m_para_buffer_RT[2], m_q_w_curr.w() = m_para_buffer_RT[3];
        for (int i = 0; i < 3; i++) m_t_w_curr(i) = m_para_buffer_RT[4 + i];
    }
    template <class Cloud>
    void upload_if_changed(Handle_pool::Shared_map &sm, int kind, const Cloud &c)
    {
        Handle_pool::Key now;
        now.n = c.points.size();
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)now.n;
        for (size_t i = 0; i < now.n; i++) {  // every point: ~1 ns each, against a grid build of milliseconds
            uint32_t b[3];
            const float v[3] = {c.points[i].x, c.points[i].y, c.points[i].z};
            std::memcpy(b, v, 12);
            h = (h ^ (((uint64_t)b[1] << 32) | b[0])) * 0xD6E8FEB86659FD93ull;
            h = (h ^ (h >> 32) ^ b[2]) * 0xFF51AFD7ED558CCDull;
        }
        now.hash = h;
        Handle_pool::Key &k = sm.key[kind];
        if (k.n == now.n && k.hash == now.hash && k.generation == ll_map_generation(sm.map, kind)) return;  // same contents, still the published structure
        int64_t gen = -1;  // the generation of OUR publication, reported from inside the map's lock
        if (now.n == 0) {
            const float none[4] = {0, 0, 0, 0};
            check(ll_map_upload_gen(sm.map, kind, none, 4, 0, 0.0f, &gen), "ll_map_upload");
        } else {
            // ll_map_upload takes x, y, z at any float stride: a point type whose coordinates are three consecutive floats
            // (pcl::PointXYZI: 8 floats per point) goes as it lies, without a staging copy
            const auto &p0 = c.points[0];
            const bool strided = sizeof(p0) % sizeof(float) == 0 && (const void *)(&p0.x + 1) == (const void *)&p0.y &&
                                 (const void *)(&p0.x + 2) == (const void *)&p0.z;
            if (strided) {
                check(ll_map_upload_gen(sm.map, kind, &p0.x, (int32_t)(sizeof(p0) / sizeof(float)), (int64_t)now.n, 0.0f, &gen), "ll_map_upload");
            } else {
                const std::vector<float> v = cloud_to_xyzi(c);
                check(ll_map_upload_gen(sm.map, kind, v.data(), 4, (int64_t)now.n, 0.0f, &gen), "ll_map_upload");
            }
        }
        now.generation = gen;  // (not ll_map_generation() now: a refresh may have published in between, and the key would then vouch for it)
        k = now;
    }
    ll_reg *reg_ = nullptr;
};

}  // namespace loam_livox_hip

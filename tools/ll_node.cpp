// ll_node -- a ROS-free, compiled host for the two node bodies of hku-mars/loam_livox that sit either side of the hot
// path, running on include/loam_livox_adapter.hpp (and through it the C ABI / the HIP library):
//
//   Laser_feature::laserCloudHandler, Livox branch      source/laser_feature_extractor.hpp:241-392
//   Laser_mapping::process_new_scan (+ the three cloud  source/laser_mapping.hpp:1316-1520, 749-780
//   handlers and init_pointcloud_registration)          source/laser_mapping.hpp:1266-1297
//   Laser_mapping::process, the queue of complete       source/laser_mapping.hpp:89-120, 633-647, 1697-1711
//   triples and the maximum_mapping_buffer drop rule
//   the node's outputs: /velodyne_cloud_registered,     source/laser_mapping.hpp:1570-1575, 1613-1653
//   /aft_mapped_to_init, /aft_mapped_path, the
//   camera_init -> aft_mapped transform
//   service_pub_surround_pts: /laser_cloud_surround     source/laser_mapping.hpp:1151-1200, 1567
//
// ROS is absent from the image, so the node/topic surface is kept in shape only: messages are PointCloud2-shaped structs
// (header, fields, point_step, byte payload), nodes talk through named in-process topics with the reference's names
// (/laser_points_<i> -> /pc2_full, /pc2_surface, /pc2_corners -> mapping), parameters come from a flat name=value table
// with the reference's parameter names.  What the reference does on service threads (update_buff_for_matching,
// laser_mapping.hpp:568-594) is done synchronously after every accepted frame, so a run is reproducible.
//
//   ll_node --in seq.bin --out log.txt [--param name=value ...] [--dump-io io.bin]
//   ll_node --replay-io io.bin --out log.txt [--param name=value ...]
//
// Outputs are message-shaped structs with the field layout and frame ids of nav_msgs/Odometry, nav_msgs/Path,
// geometry_msgs/PoseStamped, tf::StampedTransform and sensor_msgs/PointCloud2, published on the reference's topic names; a
// recorder subscribed to every output topic writes one log line per message (ODOM / PATH / TF / CLOUD).  --dump-io records
// what the mapping node was handed (handler calls, process passes, publish calls); --replay-io drives the same handler /
// queue / publish code from such a record without touching the GPU -- tests/test_ll_node_outputs.py runs the reference's
// own text (laser_mapping.hpp:89-120, 633-647, 749-780, 1701-1735, 1570-1575, 1613-1653 compiled against stub ROS types) on
// the same record and compares every field.
//
// seq.bin: "LLSEQ001", int32 n_messages, then per message { int32 lidar_index, float64 stamp, int32 n_points,
// n_points x 4 float32 (x, y, z, intensity) } -- tools/ll_sequence.py writes it.  log.txt gets one line per published
// piece (PUB: sizes and hashes of the three clouds), one per processed scan (REG: result, pose, stack and
// match-buffer sizes) and one per output message; tests/test_ll_node.py compares PUB / REG with the Python mirrors
// (feature_node.py / mapping.py).
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "loam_livox_adapter.hpp"

namespace ll = loam_livox_hip;

// ------------------------------------------------------------------------------------------------ message types
struct Header {
    uint32_t seq = 0;
    double stamp = 0;
    std::string frame_id;
};
struct PointField {
    std::string name;
    uint32_t offset = 0;
    uint8_t datatype = 7;  // FLOAT32
    uint32_t count = 1;
};
struct PointCloud2 {
    Header header;
    uint32_t height = 1, width = 0;
    std::vector<PointField> fields;
    bool is_bigendian = false;
    uint32_t point_step = 0, row_step = 0;
    std::vector<uint8_t> data;
    bool is_dense = true;
};

// geometry_msgs / nav_msgs / tf shapes (field names and nesting of the ROS messages the reference fills, laser_mapping.hpp:1613-1653)
struct Point {
    double x = 0, y = 0, z = 0;
};
struct Quaternion {
    double x = 0, y = 0, z = 0, w = 1;
};
struct Pose {
    Point position;
    Quaternion orientation;
};
struct PoseWithCovariance {
    Pose pose;
    double covariance[36] = {0};
};
struct Vector3 {
    double x = 0, y = 0, z = 0;
};
struct Twist {
    Vector3 linear, angular;
};
struct TwistWithCovariance {
    Twist twist;
    double covariance[36] = {0};
};
struct Odometry {  // nav_msgs/Odometry
    Header header;
    std::string child_frame_id;
    PoseWithCovariance pose;
    TwistWithCovariance twist;
};
struct PoseStamped {  // geometry_msgs/PoseStamped
    Header header;
    Pose pose;
};
struct Path {  // nav_msgs/Path
    Header header;
    std::vector<PoseStamped> poses;
};
struct StampedTransform {  // tf::StampedTransform( transform, stamp, frame_id, child_frame_id )
    double stamp_ = 0;
    std::string frame_id_, child_frame_id_;
    double origin[3] = {0, 0, 0};
    double rotation[4] = {0, 0, 0, 1};  // x, y, z, w
};

struct PointXYZI {
    float x = 0, y = 0, z = 0, intensity = 0;
};
struct Cloud {
    typedef std::shared_ptr<Cloud> Ptr;
    std::vector<PointXYZI> points;
    size_t size() const { return points.size(); }
    Cloud &operator+=(const Cloud &o)
    {
        points.insert(points.end(), o.points.begin(), o.points.end());
        return *this;
    }
};

// pcl::toROSMsg of a PointCloud<PointXYZI>: PCL copies its 32-byte point structs as they lie -- x, y, z at 0 / 4 / 8 (data[3] = 1.0f at 12),
// intensity at 16, padding to 32 -- and lists the four fields; subscribers decode by field offsets.
static void toROSMsg(const Cloud &c, PointCloud2 &m)
{
    static const char *names[4] = {"x", "y", "z", "intensity"};
    static const uint32_t offsets[4] = {0, 4, 8, 16};
    m.fields.resize(4);
    for (uint32_t i = 0; i < 4; i++) {
        m.fields[i].name = names[i];
        m.fields[i].offset = offsets[i];
    }
    m.height = 1;
    m.width = (uint32_t)c.size();
    m.point_step = 32;
    m.row_step = 32 * m.width;
    m.data.assign((size_t)m.row_step, 0);
    const float one = 1.0f;
    for (size_t i = 0; i < c.size(); i++) {
        uint8_t *p = m.data.data() + 32 * i;
        std::memcpy(p, &c.points[i].x, 12);
        std::memcpy(p + 12, &one, 4);
        std::memcpy(p + 16, &c.points[i].intensity, 4);
    }
}

static void fromROSMsg(const PointCloud2 &m, Cloud &c)
{
    int off[4] = {-1, -1, -1, -1};
    for (const PointField &f : m.fields) {
        if (f.datatype != 7) continue;
        if (f.name == "x") off[0] = (int)f.offset;
        if (f.name == "y") off[1] = (int)f.offset;
        if (f.name == "z") off[2] = (int)f.offset;
        if (f.name == "intensity") off[3] = (int)f.offset;
    }
    if (off[0] < 0 || off[1] < 0 || off[2] < 0) throw std::runtime_error("fromROSMsg: message has no float32 x / y / z fields");
    const size_t n = (size_t)m.width * m.height;
    c.points.assign(n, PointXYZI());
    for (size_t i = 0; i < n; i++) {
        const uint8_t *p = m.data.data() + i * m.point_step;
        std::memcpy(&c.points[i].x, p + off[0], 4);
        std::memcpy(&c.points[i].y, p + off[1], 4);
        std::memcpy(&c.points[i].z, p + off[2], 4);
        if (off[3] >= 0) std::memcpy(&c.points[i].intensity, p + off[3], 4);
    }
}

// ------------------------------------------------------------------------------------------------ topics, parameters
template <class Msg>
class Bus {
   public:
    typedef std::function<void(const Msg &)> Callback;
    void subscribe(const std::string &name, Callback cb) { subs_[name].push_back(cb); }
    void publish(const std::string &name, const Msg &m)
    {
        auto it = subs_.find(name);
        if (it == subs_.end()) return;
        for (Callback &cb : it->second) cb(m);
    }

   private:
    std::map<std::string, std::vector<Callback>> subs_;
};
class Topics {
   public:
    typedef Bus<PointCloud2>::Callback Callback;
    void subscribe(const std::string &name, Callback cb) { clouds.subscribe(name, cb); }
    void publish(const std::string &name, const PointCloud2 &m) { clouds.publish(name, m); }
    Bus<PointCloud2> clouds;
    Bus<Odometry> odometry;
    Bus<Path> paths;
    Bus<StampedTransform> tf;  // tf::TransformBroadcaster::sendTransform -> "/tf"
};

class Params {
   public:
    void set(const std::string &kv)
    {
        const size_t eq = kv.find('=');
        if (eq == std::string::npos) throw std::runtime_error("--param expects name=value, got " + kv);
        table_[kv.substr(0, eq)] = kv.substr(eq + 1);
    }
    template <class T>
    void param(const std::string &name, T &var, T def) const  // nh.param<T>( name, var, default )
    {
        auto it = table_.find(name);
        var = it == table_.end() ? def : (T)std::atof(it->second.c_str());
    }

   private:
    std::map<std::string, std::string> table_;
};

static uint64_t cloud_hash(const Cloud &c);
static uint64_t msg_hash(const PointCloud2 &m)  // of the payload as the subscriber decodes it
{
    Cloud c;
    fromROSMsg(m, c);
    return cloud_hash(c);
}

// The recorder of the node's outputs: one log line per message on every output topic, all fields.
static void record_outputs(Topics &topics, FILE *log)
{
    for (const char *name : {"/velodyne_cloud_registered", "/laser_cloud_surround"})
        topics.clouds.subscribe(name, [log, name](const PointCloud2 &m) {
            std::fprintf(log, "CLOUD %s %.17g %s %u %u %u %d %016" PRIx64 "\n", name, m.header.stamp, m.header.frame_id.c_str(), m.width, m.height, m.point_step,
                         (int)m.fields.size(), msg_hash(m));
        });
    topics.odometry.subscribe("/aft_mapped_to_init", [log](const Odometry &o) {
        std::fprintf(log, "ODOM /aft_mapped_to_init %.17g %s %s %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", o.header.stamp, o.header.frame_id.c_str(),
                     o.child_frame_id.c_str(), o.pose.pose.position.x, o.pose.pose.position.y, o.pose.pose.position.z, o.pose.pose.orientation.x,
                     o.pose.pose.orientation.y, o.pose.pose.orientation.z, o.pose.pose.orientation.w);
    });
    topics.paths.subscribe("/aft_mapped_path", [log](const Path &p) {
        const PoseStamped &l = p.poses.back();
        std::fprintf(log, "PATH /aft_mapped_path %.17g %s %zu %.17g %s %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", p.header.stamp, p.header.frame_id.c_str(),
                     p.poses.size(), l.header.stamp, l.header.frame_id.c_str(), l.pose.position.x, l.pose.position.y, l.pose.position.z, l.pose.orientation.x,
                     l.pose.orientation.y, l.pose.orientation.z, l.pose.orientation.w);
    });
    topics.tf.subscribe("/tf", [log](const StampedTransform &t) {
        std::fprintf(log, "TF %.17g %s %s %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", t.stamp_, t.frame_id_.c_str(), t.child_frame_id_.c_str(), t.origin[0], t.origin[1],
                     t.origin[2], t.rotation[0], t.rotation[1], t.rotation[2], t.rotation[3]);
    });
}

static uint64_t cloud_hash(const Cloud &c)  // position-weighted sum of the cloud's 32-bit words, mod 2^64 (tools/ll_sequence.py: cloud_hash)
{
    uint64_t h = 0;
    const uint32_t *w = (const uint32_t *)c.points.data();
    for (size_t i = 0; i < c.size() * 4; i++) h += (uint64_t)w[i] * (uint64_t)(2 * i + 1);
    return h;
}

// ------------------------------------------------------------------------------------------------ feature node
class Laser_feature {
   public:
    Laser_feature(Topics &topics, const Params &prm, FILE *log) : topics_(topics), log_(log)
    {
        // laser_feature_extractor.hpp:135-160 (parameter names as in the launch files)
        prm.param<int>("common/piecewise_number", m_piecewise_number, 3);
        prm.param<int>("common/if_motion_deblur", m_if_motion_deblur, 0);
        prm.param<int>("common/odom_mode", m_odom_mode, 0);
        prm.param<int>("common/maximum_input_lidar_pointcloud", m_maximum_input_lidar_pointcloud, 3);
        prm.param<int>("feature_extraction/system_delay", m_para_system_delay, 20);
        prm.param<float>("feature_extraction/mapping_plane_resolution", m_plane_resolution, 0.8f);
        prm.param<float>("feature_extraction/mapping_line_resolution", m_line_resolution, 0.8f);
        prm.param<float>("feature_extraction/livox_min_dis", m_livox.m_livox_min_allow_dis, 1.0f);
        prm.param<float>("feature_extraction/livox_min_sigma", m_livox.m_livox_min_sigma, 7e-3f);
        prm.param<float>("feature_extraction/corner_curvature", m_livox.thr_corner_curvature, 0.05f);
        prm.param<float>("feature_extraction/surface_curvature", m_livox.thr_surface_curvature, 0.01f);
        prm.param<float>("feature_extraction/minimum_view_angle", m_livox.minimum_view_angle, 10.0f);
        prm.param<int>("ll/max_points", m_livox.max_points, 24000);
        m_livox.piecewise_number = m_if_motion_deblur ? 1 : m_piecewise_number;
        m_voxel_filter_for_surface.setLeafSize(m_plane_resolution / 2, m_plane_resolution / 2, m_plane_resolution / 2);  // :192
        m_voxel_filter_for_corner.setLeafSize(m_line_resolution, m_line_resolution, m_line_resolution);                  // :193
        m_voxel_filter_for_surface.max_points = m_voxel_filter_for_corner.max_points = m_livox.max_points * m_maximum_input_lidar_pointcloud;
        m_map_pointcloud_full_vec_vec.assign((size_t)m_maximum_input_lidar_pointcloud, std::vector<Cloud>((size_t)m_piecewise_number));
        m_map_pointcloud_surface_vec_vec = m_map_pointcloud_corner_vec_vec = m_map_pointcloud_full_vec_vec;
        for (int i = 0; i < m_maximum_input_lidar_pointcloud; i++)  // :207-215: one subscription per lidar
            topics_.subscribe("/laser_points_" + std::to_string(i), [this, i](const PointCloud2 &m) { laserCloudHandler(m, i); });
    }

    void laserCloudHandler(const PointCloud2 &laserCloudMsg, int current_lidar_index)
    {
        if (!m_para_systemInited) {  // :258-267: the first messages of a session are dropped
            if (++m_para_system_init_count >= m_para_system_delay)
                m_para_systemInited = true;
            else
                return;
        }
        Cloud laserCloudIn;
        fromROSMsg(laserCloudMsg, laserCloudIn);
        std::vector<Cloud> laserCloudScans = m_livox.extract_laser_features(laserCloudIn, laserCloudMsg.header.stamp);  // :285
        if (laserCloudScans.size() <= 5) return;                                                                        // :287-290
        const double n_scans = (double)laserCloudScans.size();
        const int piece_wise = m_if_motion_deblur ? 1 : m_piecewise_number;  // :305-309
        std::vector<float> piece_start((size_t)piece_wise), piece_end((size_t)piece_wise);
        const float n_pts = (float)m_livox.m_pts_info_vec.size();
        for (int i = 0; i < piece_wise; i++) {  // :312-324: petal range of the piece -> blur-ratio window
            const int first_petal = (int)(n_scans * i / piece_wise), last_petal = (int)(n_scans * (i + 1) / piece_wise) - 1;
            piece_start[(size_t)i] = (float)m_livox.find_pt_info(laserCloudScans[(size_t)first_petal].points.front())->idx / n_pts;
            piece_end[(size_t)i] = (float)m_livox.find_pt_info(laserCloudScans[(size_t)last_petal].points.back())->idx / n_pts;
        }
        for (int i = 0; i < piece_wise; i++)  // :326-334
            m_livox.get_features(m_map_pointcloud_corner_vec_vec[(size_t)current_lidar_index][(size_t)i],
                                 m_map_pointcloud_surface_vec_vec[(size_t)current_lidar_index][(size_t)i],
                                 m_map_pointcloud_full_vec_vec[(size_t)current_lidar_index][(size_t)i], piece_start[(size_t)i], piece_end[(size_t)i]);
        if (current_lidar_index != 0) return;  // :348-351: only lidar 0 publishes
        for (int i = 0; i < piece_wise; i++) {
            Cloud::Ptr livox_full(new Cloud()), livox_surface(new Cloud()), livox_corners(new Cloud());
            for (int ii = 0; ii < m_maximum_input_lidar_pointcloud; ii++) {  // :353-358: the Mid-100's heads are merged here
                *livox_full += m_map_pointcloud_full_vec_vec[(size_t)ii][(size_t)i];
                *livox_surface += m_map_pointcloud_surface_vec_vec[(size_t)ii][(size_t)i];
                *livox_corners += m_map_pointcloud_corner_vec_vec[(size_t)ii][(size_t)i];
            }
            m_voxel_filter_for_surface.setInputCloud(livox_surface);  // :372-373
            m_voxel_filter_for_surface.filter(*livox_surface);
            m_voxel_filter_for_corner.setInputCloud(livox_corners);   // :379-380
            m_voxel_filter_for_corner.filter(*livox_corners);
            std::fprintf(log_, "PUB %zu %zu %zu %016" PRIx64 " %016" PRIx64 " %016" PRIx64 "\n", livox_full->size(), livox_surface->size(),
                         livox_corners->size(), cloud_hash(*livox_full), cloud_hash(*livox_surface), cloud_hash(*livox_corners));
            PointCloud2 out;
            // ros::Time::now() in the node (:340), distinct for every piece -- the mapping node pairs the three clouds of a piece by this
            // stamp (laser_mapping.hpp:633-647).  Here: the message's own stamp plus a microsecond per piece, so that runs are reproducible
            out.header.stamp = laserCloudMsg.header.stamp + 1e-6 * i;
            out.header.frame_id = "camera_init";
            toROSMsg(*livox_full, out);
            topics_.publish("/pc2_full", out);
            toROSMsg(*livox_surface, out);
            topics_.publish("/pc2_surface", out);
            toROSMsg(*livox_corners, out);
            topics_.publish("/pc2_corners", out);
            if (m_odom_mode == 0) break;  // :385-388
        }
    }

    int m_piecewise_number = 3, m_if_motion_deblur = 0, m_odom_mode = 0, m_maximum_input_lidar_pointcloud = 3;
    int m_para_system_delay = 20, m_para_system_init_count = 0;
    bool m_para_systemInited = false;
    float m_plane_resolution = 0.8f, m_line_resolution = 0.8f;
    ll::Livox_laser m_livox;  // a single extractor for every lidar: its time base runs across the messages (:92)
    ll::VoxelGrid<Cloud> m_voxel_filter_for_surface, m_voxel_filter_for_corner;
    std::vector<std::vector<Cloud>> m_map_pointcloud_full_vec_vec, m_map_pointcloud_surface_vec_vec, m_map_pointcloud_corner_vec_vec;

   private:
    Topics &topics_;
    FILE *log_;
};

// ------------------------------------------------------------------------------------------------ mapping node
// io record (--dump-io / --replay-io): "LLIO0001", then events { uint8 type; ... } --
//   'C' / 'S' / 'F'  a message on /pc2_corners, /pc2_surface, /pc2_full: float64 stamp, int32 n, n x 4 float32
//   'P'              one pass of the process loop body (laser_mapping.hpp:1701-1735)
//   'U'              a publish call: int32 frame_index, float64 time_odom, float64 pose[7], int32 n, n x 4 float32 (the registered full cloud)
struct Data_pair {  // laser_mapping.hpp:89-120
    PointCloud2 m_pc_corner, m_pc_full, m_pc_plane;
    bool m_has_pc_corner = false, m_has_pc_full = false, m_has_pc_plane = false;
    void add_pc_corner(const PointCloud2 &m) { m_pc_corner = m, m_has_pc_corner = true; }
    void add_pc_plane(const PointCloud2 &m) { m_pc_plane = m, m_has_pc_plane = true; }
    void add_pc_full(const PointCloud2 &m) { m_pc_full = m, m_has_pc_full = true; }
    bool is_completed() const { return m_has_pc_corner & m_has_pc_full & m_has_pc_plane; }
};

class Laser_mapping {
   public:
    // dry: handler / queue / publish code only -- no registration, no device (--replay-io)
    Laser_mapping(Topics &topics, const Params &prm, FILE *log, bool dry = false, FILE *io_dump = nullptr) : topics_(topics), log_(log), dry_(dry), io_(io_dump)
    {
        // laser_mapping.hpp:640-760 (parameter names as in the launch files)
        prm.param<int>("common/if_motion_deblur", m_if_motion_deblur, 0);
        prm.param<int>("mapping/init_accumulate_frames", m_mapping_init_accumulate_frames, 50);
        prm.param<int>("mapping/maximum_histroy_buffer", m_maximum_history_size, 100);
        prm.param<int>("mapping/if_input_downsample", m_if_input_downsample_mode, 1);
        prm.param<float>("mapping/mapping_line_resolution", m_line_resolution, 0.1f);
        prm.param<float>("mapping/mapping_plane_resolution", m_plane_resolution, 0.4f);
        prm.param<float>("mapping/max_allow_incre_R", m_para_max_angular_rate, 200.0f / 50.0f);
        prm.param<float>("mapping/max_allow_incre_T", m_para_max_speed, 100.0f / 50.0f);
        prm.param<float>("mapping/max_allow_final_cost", m_max_final_cost, 100.0f);
        prm.param<double>("mapping/minimum_icp_R_diff", m_minimum_icp_R_diff, 0.01);
        prm.param<double>("mapping/minimum_icp_T_diff", m_minimum_icp_T_diff, 0.01);
        prm.param<double>("mapping/history_add_t_step", m_history_add_t_step, 0.0);
        prm.param<double>("mapping/history_add_angle_step", m_history_add_angle_step, 0.0);
        prm.param<int>("optimization/icp_maximum_iteration", m_para_icp_max_iterations, 20);
        prm.param<int>("optimization/ceres_maximum_iteration", m_para_cere_max_iterations, 100);
        prm.param<int>("optimization/maximum_residual_blocks", m_para_optimization_maximum_residual_block, 100000);
        prm.param<int>("ll/subsample_seed", m_subsample_seed, 1);
        prm.param<int>("ll/max_points", m_max_points, 24000);
        m_down_sample_filter_corner.setLeafSize(m_line_resolution, m_line_resolution, m_line_resolution);     // :742
        m_down_sample_filter_surface.setLeafSize(m_plane_resolution, m_plane_resolution, m_plane_resolution);  // :743
        m_down_sample_filter_corner.max_points = m_down_sample_filter_surface.max_points = 3 * m_max_points;
        prm.param<int>("mapping/maximum_mapping_buffer", m_max_buffer_size, 5);                       // :688
        prm.param<int>("common/maximum_parallel_thread", m_maximum_parallel_thread, 2);               // :671 (scans are processed one at a time here)
        prm.param<double>("mapping/surround_pointcloud_resolution", m_surround_pointcloud_resolution, 0.5);  // :696
        prm.param<float>("mapping/pt_cell_resolution", m_pt_cell_resolution, 1.0f);                     // :690
        prm.param<int>("ll/surround_every_frames", m_surround_every_frames, 100);  // service_pub_surround_pts wakes up every 100 frames (:1163)
        prm.param<int>("ll/process_every_messages", m_process_every_messages, 1);  // > 1: the mapping loop runs only after every k-th input message (a slow mapper: exercises the drop rule)
        if (!dry_) history_.reset(new ll::History_buffer(m_maximum_history_size, 3 * m_max_points, m_line_resolution, m_plane_resolution));
        // :596-598
        topics.subscribe("/pc2_corners", [this](const PointCloud2 &m) { laserCloudCornerLastHandler(m); spin(); });
        topics.subscribe("/pc2_surface", [this](const PointCloud2 &m) { laserCloudSurfLastHandler(m); spin(); });
        topics.subscribe("/pc2_full", [this](const PointCloud2 &m) { laserCloudFullResHandler(m); spin(); });
    }

    // laser_mapping.hpp:633-647.  (The reference never erases a map entry and deletes the pair after use, :1734: a later message with the
    // stamp of a processed triple would touch freed memory.  Here a completed pair leaves the map when it is queued.)
    Data_pair *get_data_pair(const double &time_stamp)
    {
        auto it = m_map_data_pair.find(time_stamp);
        if (it == m_map_data_pair.end()) it = m_map_data_pair.insert(std::make_pair(time_stamp, std::make_shared<Data_pair>())).first;
        return it->second.get();
    }
    void queue_if_completed(double stamp)
    {
        auto it = m_map_data_pair.find(stamp);
        if (it->second->is_completed()) {
            m_queue_avail_data.push_back(it->second);
            m_map_data_pair.erase(it);
        }
    }
    // laser_mapping.hpp:749-780
    void laserCloudCornerLastHandler(const PointCloud2 &laserCloudCornerLast2)
    {
        io_message('C', laserCloudCornerLast2);
        get_data_pair(laserCloudCornerLast2.header.stamp)->add_pc_corner(laserCloudCornerLast2);
        queue_if_completed(laserCloudCornerLast2.header.stamp);
    }
    void laserCloudSurfLastHandler(const PointCloud2 &laserCloudSurfLast2)
    {
        io_message('S', laserCloudSurfLast2);
        get_data_pair(laserCloudSurfLast2.header.stamp)->add_pc_plane(laserCloudSurfLast2);
        queue_if_completed(laserCloudSurfLast2.header.stamp);
    }
    void laserCloudFullResHandler(const PointCloud2 &laserCloudFullRes2)
    {
        io_message('F', laserCloudFullRes2);
        get_data_pair(laserCloudFullRes2.header.stamp)->add_pc_full(laserCloudFullRes2);
        queue_if_completed(laserCloudFullRes2.header.stamp);
    }
    // the body of Laser_mapping::process's loop runs while triples are waiting; with ll/process_every_messages = k only after every
    // k-th input message (the reference's loop runs beside the handlers and falls behind when registration is slower than the lidar)
    void spin()
    {
        if (++m_messages_seen % m_process_every_messages != 0) return;
        while (!m_queue_avail_data.empty()) process_once();
    }
    // one pass of the loop body, laser_mapping.hpp:1697-1742
    void process_once()
    {
        if (m_queue_avail_data.empty()) return;  // (the reference sleeps until a triple is there, :1697-1700)
        if (io_) std::fputc('P', io_);
        while (m_queue_avail_data.size() >= (unsigned int)m_max_buffer_size) {  // :1702-1707 "Drop lidar frame in mapping for real time performance !!!"
            std::fprintf(log_, "DROP %.17g\n", m_queue_avail_data.front()->m_pc_corner.header.stamp);
            m_queue_avail_data.pop_front();
            if (m_queue_avail_data.empty()) return;  // (maximum_mapping_buffer <= 1: the reference would read the front of an empty queue)
        }
        std::shared_ptr<Data_pair> current_data_pair = m_queue_avail_data.front();
        m_queue_avail_data.pop_front();
        m_time_pc_corner_past = current_data_pair->m_pc_corner.header.stamp;  // :1715
        fromROSMsg(current_data_pair->m_pc_corner, m_laser_cloud_corner_last);  // :1725-1732
        fromROSMsg(current_data_pair->m_pc_plane, m_laser_cloud_surf_last);
        fromROSMsg(current_data_pair->m_pc_full, m_laser_cloud_full_res);
        std::fprintf(log_, "TAKE %.17g %.17g %.17g %zu %zu %zu %016" PRIx64 " %016" PRIx64 " %016" PRIx64 "\n", current_data_pair->m_pc_corner.header.stamp,
                     current_data_pair->m_pc_plane.header.stamp, current_data_pair->m_pc_full.header.stamp, m_laser_cloud_corner_last.size(),
                     m_laser_cloud_surf_last.size(), m_laser_cloud_full_res.size(), cloud_hash(m_laser_cloud_corner_last), cloud_hash(m_laser_cloud_surf_last),
                     cloud_hash(m_laser_cloud_full_res));
        // :1737-1742 hands the scan to one of maximum_parallel_thread asynchronous tasks; here it is processed before the next one is taken
        if (!dry_) process_new_scan();
    }

    // what process_new_scan publishes after a successful registration, laser_mapping.hpp:1570-1575 and 1613-1653
    void publish_frame(const Cloud &current_laser_cloud_full, double time_odom, const double pose[7])
    {
        if (io_) {
            std::fputc('U', io_);
            const int32_t fi = m_current_frame_index, n = (int32_t)current_laser_cloud_full.size();
            std::fwrite(&fi, 4, 1, io_);
            std::fwrite(&time_odom, 8, 1, io_);
            std::fwrite(pose, 8, 7, io_);
            std::fwrite(&n, 4, 1, io_);
            std::fwrite(current_laser_cloud_full.points.data(), sizeof(PointXYZI), (size_t)n, io_);
        }
        PointCloud2 laserCloudFullRes3;  // :1571-1575
        toROSMsg(current_laser_cloud_full, laserCloudFullRes3);
        laserCloudFullRes3.header.stamp = time_odom;
        laserCloudFullRes3.header.frame_id = "camera_init";
        topics_.clouds.publish("/velodyne_cloud_registered", laserCloudFullRes3);  // single_frame_with_pose_tranfromed

        Odometry odomAftMapped;  // :1615-1629
        odomAftMapped.header.frame_id = "camera_init";
        odomAftMapped.child_frame_id = "aft_mapped";
        odomAftMapped.header.stamp = time_odom;
        odomAftMapped.pose.pose.orientation.x = pose[0];
        odomAftMapped.pose.pose.orientation.y = pose[1];
        odomAftMapped.pose.pose.orientation.z = pose[2];
        odomAftMapped.pose.pose.orientation.w = pose[3];
        odomAftMapped.pose.pose.position.x = pose[4];
        odomAftMapped.pose.pose.position.y = pose[5];
        odomAftMapped.pose.pose.position.z = pose[6];
        topics_.odometry.publish("/aft_mapped_to_init", odomAftMapped);

        PoseStamped pose_aft_mapped;  // :1631-1641
        pose_aft_mapped.header = odomAftMapped.header;
        pose_aft_mapped.pose = odomAftMapped.pose.pose;
        m_laser_after_mapped_path.header.stamp = odomAftMapped.header.stamp;
        m_laser_after_mapped_path.header.frame_id = "camera_init";
        if (m_current_frame_index % 10 == 0) {
            m_laser_after_mapped_path.poses.push_back(pose_aft_mapped);
            topics_.paths.publish("/aft_mapped_path", m_laser_after_mapped_path);
        }

        StampedTransform transform;  // :1643-1653
        transform.origin[0] = pose[4], transform.origin[1] = pose[5], transform.origin[2] = pose[6];
        transform.rotation[3] = pose[3];
        transform.rotation[0] = pose[0];
        transform.rotation[1] = pose[1];
        transform.rotation[2] = pose[2];
        transform.stamp_ = odomAftMapped.header.stamp;
        transform.frame_id_ = "camera_init";
        transform.child_frame_id_ = "aft_mapped";
        topics_.tf.publish("/tf", transform);
    }

    // service_pub_surround_pts, laser_mapping.hpp:1151-1200: the reference's service thread wakes up when the frame counter has advanced by
    // 100 and publishes the full-cloud cell map around the current position, every cell down-sampled, then the union.  Here it runs at the end
    // of the frame that makes the counter reach the step (stamp: the frame's time_odom instead of ros::Time::now()).
    void pub_surround_pts(double stamp)
    {
        if (m_current_frame_index - m_surround_last_update_index < m_surround_every_frames) return;  // :1163
        m_surround_last_update_index = m_current_frame_index;
        if (!m_pt_cell_map_full || m_pt_cell_map_full->get_cells_size() == 0) return;  // :1170
        Cloud::Ptr laser_cloud_surround(new Cloud());
        // find_cells_in_radius( m_t_w_curr, 1000.0 ) (:1172), each cell through the voxel filter (:1175-1182; m_down_sample_replace = 1, :277; the
        // filtered cloud is NOT stored back, :1180), concatenated
        m_pt_cell_map_full->find_cells_in_radius_filtered(pose_, 1000.0f, (float)m_surround_pointcloud_resolution, *laser_cloud_surround);
        if (laser_cloud_surround->size()) {  // :1189-1197
            m_down_sample_filter_surround.setLeafSize((float)m_surround_pointcloud_resolution, (float)m_surround_pointcloud_resolution, (float)m_surround_pointcloud_resolution);
            m_down_sample_filter_surround.setInputCloud(laser_cloud_surround);
            m_down_sample_filter_surround.filter(*laser_cloud_surround);
            PointCloud2 ros_laser_cloud_surround;
            toROSMsg(*laser_cloud_surround, ros_laser_cloud_surround);
            ros_laser_cloud_surround.header.stamp = stamp;
            ros_laser_cloud_surround.header.frame_id = "camera_init";
            topics_.clouds.publish("/laser_cloud_surround", ros_laser_cloud_surround);
        }
    }

    // Laser_mapping::init_pointcloud_registration, laser_mapping.hpp:1266-1297
    void init_pointcloud_registration(ll::Point_cloud_registration &pc_reg)
    {
        pc_reg.m_if_motion_deblur = m_if_motion_deblur;
        pc_reg.m_current_frame_index = m_current_frame_index;
        pc_reg.m_mapping_init_accumulate_frames = m_mapping_init_accumulate_frames;
        pc_reg.m_last_time_stamp = m_last_time_stamp;
        pc_reg.m_para_max_angular_rate = m_para_max_angular_rate;
        pc_reg.m_para_max_speed = m_para_max_speed;
        pc_reg.m_max_final_cost = m_max_final_cost;
        pc_reg.m_para_icp_max_iterations = m_para_icp_max_iterations;
        pc_reg.m_para_cere_max_iterations = m_para_cere_max_iterations;
        pc_reg.m_maximum_allow_residual_block = m_para_optimization_maximum_residual_block;
        pc_reg.m_subsample_seed = m_para_optimization_maximum_residual_block < 3 * m_max_points ? m_subsample_seed : 0;
        pc_reg.m_minimum_pt_time_stamp = m_minimum_pt_time_stamp;
        pc_reg.m_maximum_pt_time_stamp = m_maximum_pt_time_stamp;
        pc_reg.m_minimum_icp_R_diff = m_minimum_icp_R_diff;
        pc_reg.m_minimum_icp_T_diff = m_minimum_icp_T_diff;
        pc_reg.max_features = 3 * m_max_points;
        for (int i = 0; i < 4; i++) pc_reg.m_para_buffer_RT[i] = pose_[i];
        pc_reg.m_q_w_curr.x() = pose_[0], pc_reg.m_q_w_curr.y() = pose_[1], pc_reg.m_q_w_curr.z() = pose_[2], pc_reg.m_q_w_curr.w() = pose_[3];
        pc_reg.m_q_w_last = pc_reg.m_q_w_curr;
        for (int i = 0; i < 3; i++) pc_reg.m_t_w_curr(i) = pc_reg.m_t_w_last(i) = pose_[4 + i];
    }

    // Laser_mapping::process_new_scan, laser_mapping.hpp:1316-1520
    int process_new_scan()
    {
        float min_t = 0, max_t = 0;  // find_min_max_intensity( full ): the full cloud carries the time stamps (:1336)
        for (size_t i = 0; i < m_laser_cloud_full_res.size(); i++) {
            const float t = m_laser_cloud_full_res.points[i].intensity;
            if (i == 0 || t < min_t) min_t = t;
            if (i == 0 || t > max_t) max_t = t;
        }
        m_minimum_pt_time_stamp = m_last_time_stamp;  // :1345-1347
        m_maximum_pt_time_stamp = max_t;
        m_last_time_stamp = max_t;
        ll::Point_cloud_registration pc_reg;  // a fresh registrar per scan, like the node (:1348); its device handle is pooled
        init_pointcloud_registration(pc_reg);
        m_current_frame_index++;
        Cloud::Ptr laserCloudCornerStack(new Cloud()), laserCloudSurfStack(new Cloud());
        if (m_if_input_downsample_mode) {  // :1367-1373
            m_down_sample_filter_corner.setInputCloud(&m_laser_cloud_corner_last);
            m_down_sample_filter_corner.filter(*laserCloudCornerStack);
            m_down_sample_filter_surface.setInputCloud(&m_laser_cloud_surf_last);
            m_down_sample_filter_surface.filter(*laserCloudSurfStack);
        } else {
            *laserCloudCornerStack = m_laser_cloud_corner_last;
            *laserCloudSurfStack = m_laser_cloud_surf_last;
        }
        // the match buffer is resident behind pc_reg.map(): the 2-argument form registers against it (:1405-1411)
        const int reg_res = pc_reg.find_out_incremental_transfrom(laserCloudCornerStack, laserCloudSurfStack);
        int64_t n_map[2] = {map_sizes_[0], map_sizes_[1]};
        if (reg_res != 0) {  // :1413-1416 return on failure; otherwise "Add new frame" (:1417-1478) and the pose hand-over (:1496-1500)
            history_->set_gate_pose(pose_);  // m_q_w_curr / m_t_w_curr still hold the previous pose at :1439-1451
            history_->add(*laserCloudCornerStack, *laserCloudSurfStack, pc_reg.m_para_buffer_RT, m_history_add_t_step, m_history_add_angle_step);
            for (int i = 0; i < 7; i++) pose_[i] = pc_reg.m_para_buffer_RT[i];
            history_->refresh(pc_reg.map(), &n_map[0], &n_map[1]);  // update_buff_for_matching (:460-566), synchronous here
            map_sizes_[0] = n_map[0], map_sizes_[1] = n_map[1];
            // :1442 the full-resolution cloud into the map frame with the new pose, :1567 into the full-cloud cell map, :1570-1653 the outputs
            Cloud current_laser_cloud_full;
            pc_reg.pointcloudAssociateToMap(m_laser_cloud_full_res, current_laser_cloud_full, 0);
            if (!m_pt_cell_map_full) m_pt_cell_map_full.reset(new ll::Points_cloud_map((int64_t)m_max_points * 3 * 8, m_pt_cell_resolution));
            m_pt_cell_map_full->reserve_for(current_laser_cloud_full.size());
            m_pt_cell_map_full->append_cloud(current_laser_cloud_full);
            const double time_odom = m_time_pc_corner_past;  // ros::Time::now() at :1351; the triple's stamp keeps runs reproducible
            publish_frame(current_laser_cloud_full, time_odom, pose_);
            pub_surround_pts(time_odom);
        }
        std::fprintf(log_, "REG %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %zu %zu %" PRId64 " %" PRId64 " %d\n", m_current_frame_index - 1, reg_res,
                     pose_[0], pose_[1], pose_[2], pose_[3], pose_[4], pose_[5], pose_[6], laserCloudCornerStack->size(), laserCloudSurfStack->size(),
                     n_map[0], n_map[1], pc_reg.summary.icp_iterations);
        return reg_res;
    }

    int m_if_motion_deblur = 0, m_current_frame_index = 0, m_mapping_init_accumulate_frames = 50, m_maximum_history_size = 100;
    int m_if_input_downsample_mode = 1, m_para_icp_max_iterations = 20, m_para_cere_max_iterations = 100;
    int m_para_optimization_maximum_residual_block = 100000, m_subsample_seed = 1, m_max_points = 24000;
    float m_line_resolution = 0.1f, m_plane_resolution = 0.4f, m_para_max_angular_rate = 4.0f, m_para_max_speed = 2.0f, m_max_final_cost = 100.0f;
    float m_last_time_stamp = 0, m_minimum_pt_time_stamp = 0, m_maximum_pt_time_stamp = 1.0f;
    double m_minimum_icp_R_diff = 0.01, m_minimum_icp_T_diff = 0.01, m_history_add_t_step = 0.0, m_history_add_angle_step = 0.0;
    Cloud m_laser_cloud_corner_last, m_laser_cloud_surf_last, m_laser_cloud_full_res;
    int m_max_buffer_size = 5, m_maximum_parallel_thread = 2, m_surround_every_frames = 100, m_surround_last_update_index = 0, m_process_every_messages = 1;
    long m_messages_seen = 0;
    double m_surround_pointcloud_resolution = 0.5, m_time_pc_corner_past = 0;
    float m_pt_cell_resolution = 1.0f;
    std::map<double, std::shared_ptr<Data_pair>> m_map_data_pair;  // :215
    std::deque<std::shared_ptr<Data_pair>> m_queue_avail_data;    // :216
    Path m_laser_after_mapped_path;
    ll::VoxelGrid<Cloud> m_down_sample_filter_corner, m_down_sample_filter_surface, m_down_sample_filter_surround;
    std::unique_ptr<ll::Points_cloud_map> m_pt_cell_map_full;  // :276

   private:
    void io_message(char type, const PointCloud2 &m)
    {
        if (!io_) return;
        Cloud c;
        fromROSMsg(m, c);
        const int32_t n = (int32_t)c.size();
        std::fputc(type, io_);
        std::fwrite(&m.header.stamp, 8, 1, io_);
        std::fwrite(&n, 4, 1, io_);
        std::fwrite(c.points.data(), sizeof(PointXYZI), (size_t)n, io_);
    }
    Topics &topics_;
    FILE *log_;
    bool dry_ = false;
    FILE *io_ = nullptr;
    std::unique_ptr<ll::History_buffer> history_;
    double pose_[7] = {0, 0, 0, 1, 0, 0, 0};  // m_q_w_curr / m_t_w_curr
    int64_t map_sizes_[2] = {0, 0};
};

// ------------------------------------------------------------------------------------------------ driver
static bool read_exact(FILE *f, void *p, size_t n) { return n == 0 || std::fread(p, 1, n, f) == n; }

static void read_cloud(FILE *f, Cloud &c)
{
    int32_t n = 0;
    if (!read_exact(f, &n, 4) || n < 0) throw std::runtime_error("truncated io record");
    c.points.resize((size_t)n);
    if (!read_exact(f, c.points.data(), (size_t)n * sizeof(PointXYZI))) throw std::runtime_error("truncated io record");
}

// --replay-io: the mapping node's handlers, queue and publish code driven from a record, no device
static void replay_io(FILE *fi, Topics &topics, Laser_mapping &node)
{
    (void)topics;
    char magic[8];
    if (!read_exact(fi, magic, 8) || std::memcmp(magic, "LLIO0001", 8) != 0) throw std::runtime_error("not an LLIO0001 file");
    for (int type = std::fgetc(fi); type != EOF; type = std::fgetc(fi)) {
        if (type == 'C' || type == 'S' || type == 'F') {
            double stamp = 0;
            if (!read_exact(fi, &stamp, 8)) throw std::runtime_error("truncated io record");
            Cloud c;
            read_cloud(fi, c);
            PointCloud2 m;
            toROSMsg(c, m);
            m.header.stamp = stamp;
            m.header.frame_id = "camera_init";
            if (type == 'C') node.laserCloudCornerLastHandler(m);
            if (type == 'S') node.laserCloudSurfLastHandler(m);
            if (type == 'F') node.laserCloudFullResHandler(m);
        } else if (type == 'P') {
            node.process_once();
        } else if (type == 'U') {
            int32_t frame_index = 0;
            double time_odom = 0, pose[7];
            if (!read_exact(fi, &frame_index, 4) || !read_exact(fi, &time_odom, 8) || !read_exact(fi, pose, 56)) throw std::runtime_error("truncated io record");
            Cloud c;
            read_cloud(fi, c);
            node.m_current_frame_index = frame_index;
            node.publish_frame(c, time_odom, pose);
        } else {
            throw std::runtime_error("unknown event in io record");
        }
    }
}

int main(int argc, char **argv)
{
    static const char *usage = "usage: ll_node --in seq.bin --out log.txt [--param name=value ...] [--dump-io io.bin]\n"
                               "       ll_node --replay-io io.bin --out log.txt [--param name=value ...]\n";
    std::string in, out, dump_io, replay;
    Params prm;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--in" && i + 1 < argc) in = argv[++i];
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--param" && i + 1 < argc) prm.set(argv[++i]);
        else if (a == "--dump-io" && i + 1 < argc) dump_io = argv[++i];
        else if (a == "--replay-io" && i + 1 < argc) replay = argv[++i];
        else {
            std::fputs(usage, stderr);
            return 2;
        }
    }
    if ((in.empty() == replay.empty()) || out.empty()) {
        std::fputs(usage, stderr);
        return 2;
    }
    const std::string &src = in.empty() ? replay : in;
    FILE *fi = std::fopen(src.c_str(), "rb"), *fo = std::fopen(out.c_str(), "w");
    FILE *fd = dump_io.empty() ? nullptr : std::fopen(dump_io.c_str(), "wb");
    if (!fi || !fo || (!dump_io.empty() && !fd)) {
        std::fprintf(stderr, "ll_node: cannot open %s\n", !fi ? src.c_str() : (!fo ? out.c_str() : dump_io.c_str()));
        return 1;
    }
    try {
        Topics topics;
        record_outputs(topics, fo);
        if (!replay.empty()) {
            Laser_mapping mapping_node(topics, prm, fo, true);
            replay_io(fi, topics, mapping_node);
        } else {
            char magic[8];
            int32_t n_msgs = 0;
            if (!read_exact(fi, magic, 8) || std::memcmp(magic, "LLSEQ001", 8) != 0 || !read_exact(fi, &n_msgs, 4)) throw std::runtime_error("not an LLSEQ001 file");
            if (fd) std::fwrite("LLIO0001", 1, 8, fd);
            Laser_feature feature_node(topics, prm, fo);
            Laser_mapping mapping_node(topics, prm, fo, false, fd);
            for (int32_t k = 0; k < n_msgs; k++) {
                int32_t lidar = 0, n = 0;
                double stamp = 0;
                if (!read_exact(fi, &lidar, 4) || !read_exact(fi, &stamp, 8) || !read_exact(fi, &n, 4) || n < 0) throw std::runtime_error("truncated sequence file");
                Cloud c;
                c.points.resize((size_t)n);
                if (!read_exact(fi, c.points.data(), (size_t)n * sizeof(PointXYZI))) throw std::runtime_error("truncated sequence file");
                PointCloud2 m;
                toROSMsg(c, m);
                m.header.seq = (uint32_t)k;
                m.header.stamp = stamp;
                m.header.frame_id = "livox";
                topics.publish("/laser_points_" + std::to_string(lidar), m);
            }
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "ll_node: %s\n", e.what());
        std::fclose(fo);
        if (fd) std::fclose(fd);
        return 1;
    }
    std::fclose(fi);
    std::fclose(fo);
    if (fd) std::fclose(fd);
    return 0;
}

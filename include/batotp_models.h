/*
 * batotp_models.h -- built-in serial-chain model tables (data only) for batotp_hip_set_serial_model.
 *
 * The reference hard-codes its robots in Robot::set_robotType (batotp/robot.cpp:51-62) and has a dynamics
 * model for one serial robot only, the two-link arm (Robot::dynRR, robot.cpp:377-431).  BASELINE config 3
 * asks for the KUKA LWR IV+ with torque limits; its kinematic chain is the one Robot::fwdKinKuka uses
 * (robot.cpp:105-165: link lengths a0 = .3105, a1 = .4, a2 = .39, Q12 = Rz(t1) Ry(-t2), Q34 = Rz(t3) Ry(t4),
 * Q567 = Rz(t5) Ry(-t6) Rz(t7), tool point (0, -.08, .545) in the flange frame).  The inertial parameters are
 * NOMINAL values of the order published for this arm (moving mass ~14.5 kg) plus a 2.3 kg gripper lumped
 * into the last link; they are not an identified model and nothing in the reference pins them.
 */
#ifndef BATOTP_MODELS_H
#define BATOTP_MODELS_H

#include <string.h>
#include "batotp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

static inline void batotp_model_link(batotp_serial_link *L, double ax, double ay, double az, double ox, double oy, double oz,
                                     double cx, double cy, double cz, double mass, double ixx, double iyy, double izz, double fv)
{
    memset(L, 0, sizeof(*L));
    L->axis[0] = ax; L->axis[1] = ay; L->axis[2] = az;
    L->off[0] = ox; L->off[1] = oy; L->off[2] = oz;
    L->com[0] = cx; L->com[1] = cy; L->com[2] = cz;
    L->mass = mass;
    L->inertia[0] = ixx; L->inertia[1] = iyy; L->inertia[2] = izz;
    L->fv = fv;
}

/* Fills *m for BATOTP_ROBOT_KUKA (7 links, joint values in degrees, z up) or BATOTP_ROBOT_RR (the two point
 * masses of Robot::dynRR as a chain: used to check the recursion against that closed form).
 * Returns 0, or -1 for a robot type without a table. */
static inline int batotp_builtin_serial_model(int robot_type, batotp_serial_model *m)
{
    memset(m, 0, sizeof(*m));
    if (robot_type == BATOTP_ROBOT_KUKA)
    {
        m->n_links = 7;
        m->degrees = 1;                     /* robot.cpp:130-136 */
        m->gravity[2] = -9.81;              /* config.h:30 */
        /*                 axis          offset from parent joint   centre of mass        mass   Ixx     Iyy     Izz     fv  */
        batotp_model_link(&m->link[0], 0, 0, 1,   0, 0, 0,        0, -0.020, 0.200,   2.70, 0.0160, 0.0160, 0.0050, 1.0);
        batotp_model_link(&m->link[1], 0, -1, 0,  0, 0, 0.3105,   0, 0.016, 0.070,    2.70, 0.0160, 0.0160, 0.0050, 1.0);
        batotp_model_link(&m->link[2], 0, 0, 1,   0, 0, 0.2,      0, 0.020, 0.130,    2.70, 0.0160, 0.0160, 0.0050, 0.8);
        batotp_model_link(&m->link[3], 0, 1, 0,   0, 0, 0.2,      0, -0.016, 0.070,   2.70, 0.0160, 0.0160, 0.0050, 0.8);
        batotp_model_link(&m->link[4], 0, 0, 1,   0, 0, 0.2,      0, -0.020, 0.120,   1.70, 0.0098, 0.0090, 0.0035, 0.5);
        batotp_model_link(&m->link[5], 0, -1, 0,  0, 0, 0.19,     0, 0.005, 0.000,    1.60, 0.0030, 0.0030, 0.0030, 0.3);
        batotp_model_link(&m->link[6], 0, 0, 1,   0, 0, 0,        0, -0.020, 0.200,   2.60, 0.0200, 0.0200, 0.0040, 0.2);
        return 0;
    }
    if (robot_type == BATOTP_ROBOT_RR)
    {
        /* robot.cpp:388: A1 = .4, A2 = .6, m1 = 4, m2 = 8, point masses at mid-link; friction 10 (robot.cpp:423-424);
         * planar in x-y, gravity along -y (robot.cpp:426-427) */
        m->n_links = 2;
        m->degrees = 1;
        m->gravity[1] = -9.81;
        batotp_model_link(&m->link[0], 0, 0, 1,   0, 0, 0,     0.2, 0, 0,   4.0, 0, 0, 0, 10.0);
        batotp_model_link(&m->link[1], 0, 0, 1,   0.4, 0, 0,   0.3, 0, 0,   8.0, 0, 0, 0, 10.0);
        return 0;
    }
    return -1;
}

#ifdef __cplusplus
}
#endif
#endif /* BATOTP_MODELS_H */

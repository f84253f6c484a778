// matrix4x4.hpp -- host 4x4 affine matrices of the scene-preparation step (reference: matrix4x4.hpp:3-56).
//
// Same type name, field (`float data[4][4]`, row-major) and factory names as the reference, so driver code
// (main.cu:68-70) compiles unchanged.  All arithmetic is fp32, as there: the bunny's placement in every fixture
// depends on these roundings (SURVEY.md Appendix C pins the composite's rows).
#ifndef RTCUDA_HOST_MATRIX4X4_HPP
#define RTCUDA_HOST_MATRIX4X4_HPP

#include <cmath>

struct Matrix4x4 {
    Matrix4x4() {}
    Matrix4x4(float e00, float e01, float e02, float e03, float e10, float e11, float e12, float e13, float e20, float e21,
              float e22, float e23, float e30, float e31, float e32, float e33) {
        const float e[16] = {e00, e01, e02, e03, e10, e11, e12, e13, e20, e21, e22, e23, e30, e31, e32, e33};
        for (int k = 0; k < 16; k++) data[k / 4][k % 4] = e[k];
    }
    static Matrix4x4 Identity() { return Diagonal(1.f, 1.f, 1.f); }
    // matrix4x4.hpp:22-27
    static Matrix4x4 Translate(float dx, float dy, float dz) {
        Matrix4x4 m = Identity();
        m.data[0][3] = dx;
        m.data[1][3] = dy;
        m.data[2][3] = dz;
        return m;
    }
    // matrix4x4.hpp:29-34
    static Matrix4x4 Scale(float sx, float sy, float sz) { return Diagonal(sx, sy, sz); }
    // matrix4x4.hpp:36-56: rotation by theta_rad about the UNIT axis (x, y, z) -- Rodrigues' formula,
    // R = c I + (1 - c) a a^T + s [a]x, each entry evaluated in fp32 as  a_i a_j (1 - c)  +/-  a_k s
    static Matrix4x4 Rotate(float axis_x, float axis_y, float axis_z, float theta_rad) {
        const float a[3] = {axis_x, axis_y, axis_z};
        const float c = cosf(theta_rad), s = sinf(theta_rad), omc = 1.f - c;
        Matrix4x4 m = Identity();
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                if (i == j) {
                    m.data[i][j] = c + a[i] * a[i] * omc;
                } else {
                    const int k = 3 - i - j;                              // the remaining axis
                    const bool plus = (j == (i + 2) % 3);                 // sign of [a]x at (i, j): (0,2) (1,0) (2,1) positive
                    const float sym = a[i < j ? i : j] * a[i < j ? j : i] * omc;  // x*y, x*z, y*z: smaller index first
                    m.data[i][j] = plus ? sym + a[k] * s : sym - a[k] * s;
                }
            }
        return m;
    }
    float data[4][4];

private:
    static Matrix4x4 Diagonal(float a, float b, float c) {
        Matrix4x4 m;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) m.data[i][j] = 0.f;
        m.data[0][0] = a;
        m.data[1][1] = b;
        m.data[2][2] = c;
        m.data[3][3] = 1.f;
        return m;
    }
};

#endif  // RTCUDA_HOST_MATRIX4X4_HPP

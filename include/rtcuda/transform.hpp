// transform.hpp -- host Transform of the scene-preparation step (reference: transform.hpp:4-33).
//
// composite(other): matrix <- other . matrix, accumulated in fp32 from 0 in k order (transform.hpp:13-24).
// apply(v): the reference's mixed rounding, which decides the last bit of every bunny vertex: the products are
// float x double = double, summed left to right in double; x and y are then rounded to float, z stays double until
// the caller narrows it into a Vec3 (transform.hpp:26-33).
#ifndef RTCUDA_HOST_TRANSFORM_HPP
#define RTCUDA_HOST_TRANSFORM_HPP

#include <array>

#include "matrix4x4.hpp"

struct Transform {
    Transform(const Matrix4x4 &matrix) : matrix(matrix) {}

    void composite(const Matrix4x4 &other) {
        Matrix4x4 r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                float acc = 0;
                for (int k = 0; k < 4; k++) acc += other.data[i][k] * matrix.data[k][j];
                r.data[i][j] = acc;
            }
        matrix = r;
    }

    void apply(std::array<double, 3> &v) const {
        double out[3];
        for (int i = 0; i < 3; i++) {
            const float *row = matrix.data[i];
            out[i] = row[0] * v[0] + row[1] * v[1] + row[2] * v[2] + row[3];
        }
        v[0] = (float)out[0];
        v[1] = (float)out[1];
        v[2] = out[2];
    }

    Matrix4x4 matrix;
};

#endif  // RTCUDA_HOST_TRANSFORM_HPP

"""humanrf_amd: MI355X-native (gfx950) implementation of HumanRF's volumetric ray-marching hot path
(occupancy-grid ray sampling -> 4D-decomposed hash encoding -> fused density / colour MLPs -> alpha
compositing, forward and training backward) behind the reference's own operator surface:

    humanrf_amd.dataset.ray_sampler_native / occupancy_grid_native   <-> actorshq/dataset/native/*.cu
    humanrf_amd.dataset.input_batch.InputBatch                       <-> actorshq/dataset/input_batch.py
    humanrf_amd.scene_representation.{HumanRF, Decomposition4D, QueryInput, QueryOutput,
                                      tensor_composition_native}     <-> humanrf/scene_representation/*
    humanrf_amd.volume_rendering.{prune_samples, render, RenderOutput} <-> humanrf/volume_rendering.py
    humanrf_amd.input.merge_input_batches                            <-> humanrf/input.py

All compute goes through libhrf_hip.so (C ABI: include/hrf.h). There is no CPU fallback.
"""
__version__ = "0.1.0"

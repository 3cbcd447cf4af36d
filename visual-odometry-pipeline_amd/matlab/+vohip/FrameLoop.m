classdef FrameLoop < handle
    % vohip.FrameLoop -- the body of the continuous-operation loop (main_demo.m:377-427, is_BA = false)
    % as ONE library call per frame, with S.P, S.X, S.C, S.F, S.T resident on the GPU.
    %
    %   loop = vohip.FrameLoop(K, size(img0), hyper_paras, feature_params);
    %   loop.reset(img0, S.X(1:3, :), S.P, S.T_W_C);            % state after the bootstrap, main_demo.m:268-303
    %   for i = ...
    %       [T_W_C, info] = loop.step(images{i}, images{i + 1}); % the following image is optional
    %       S.est_trans(:, end + 1) = T_W_C(1:3, 4);  ...        % main_demo.m:401-403
    %       if hyper_paras.is_BA, S = loop.bundle_adjust(S, T_W_C); end   % main_BA.m:288-291
    %   end
    %   S = loop.state(S);                                       % S.P, S.X, S.C, S.F, S.T back on the host
    %
    % Random numbers: the reference draws its RANSAC samples from the global stream inside
    % ransacLocalization (datasample) and estimateFundamentalMatrix.  Here the loop draws uniforms
    % with rand -- 3 per localisation hypothesis from the global stream, rewound so that it ends
    % advanced by exactly the iterations the adaptive rule executed, and 8 per fundamental-matrix
    % trial from a stream of its own (MapStream) -- and the library maps them to index sets on the
    % device, where the set sizes are known.
    properties
        MapStream            % RandStream of the fundamental-matrix trials
        Trials = 500         % update_landmarks.m:13-14
        HaveCandidates = false
        K                    % camera matrix
    end
    methods
        function obj = FrameLoop(K, image_size, hyper_paras, feature_params)
            p = struct('r_discard_redundant', hyper_paras.r_discard_redundant, ...
                       'angle_threshold_deg', hyper_paras.angle_threshold, ...
                       'min_depth', hyper_paras.min_depth, 'max_depth', hyper_paras.max_depth, ...
                       'refine_pose', double(hyper_paras.is_refine_pose));
            if nargin > 3 && isfield(feature_params, 'MinQuality'), p.min_quality = feature_params.MinQuality; end
            if nargin > 3 && isfield(feature_params, 'FilterSize'), p.filter_size = feature_params.FilterSize; end
            if nargin > 3 && isfield(feature_params, 'MinContrast'), p.min_contrast = feature_params.MinContrast; end
            if nargin > 3 && isfield(feature_params, 'MetricThreshold'), p.metric_threshold = feature_params.MetricThreshold; end
            if isfield(hyper_paras, 'feature_extract')   % genKeypoints.m:17-19
                switch hyper_paras.feature_extract
                    case 'Harris',   p.detector = 0;
                    case 'MinEigen', p.detector = 1;
                    case 'FAST',     p.detector = 2;
                        if ~isfield(p, 'min_quality'), p.min_quality = 0.1; end   % detectFASTFeatures' default
                    case 'SURF',     p.detector = 3;                              % genKeypoints.m:16
                    case 'BRISK',    p.detector = 4;                              % genKeypoints.m:20
                        if ~isfield(p, 'min_quality'), p.min_quality = 0.1; end   % detectBRISKFeatures' default
                        if nargin > 3 && isfield(feature_params, 'NumOctaves'), p.brisk_octaves = feature_params.NumOctaves; end
                    otherwise, error('vohip:FrameLoop', 'detector %s is not built natively', hyper_paras.feature_extract);
                end
            end
            if isfield(hyper_paras, 'is_BA'), p.is_BA = double(hyper_paras.is_BA); end
            vohip_mex('frame_create', K, image_size(1), image_size(2), p);
            obj.K = double(K);
            obj.MapStream = RandStream('mt19937ar', 'Seed', 'shuffle');
        end
        function reset(obj, image, X, P, T_W_C, window_size, keyframe_d, hyper_paras)
            vohip_mex('frame_reset', image, double(X), double(P), double(T_W_C));
            obj.HaveCandidates = false;
            if nargin > 5    % with bundle adjustment: the window struct B of main_demo.m:288-303, kept by the library
                w = struct('window_size', window_size, 'keyframe_d', keyframe_d, ...
                           'min_depth', hyper_paras.min_depth, 'max_depth', hyper_paras.max_depth);
                vohip_mex('window_create', obj.K, w);
                vohip_mex('window_reset', double(X), double(P), double(T_W_C));
            end
        end
        function S = bundle_adjust(~, S, T_W_C)
            % VO_bundle_adjust.m:1-98 on the device-resident state; rewrites the poses of the optimised
            % window in S.est_rot / S.est_trans (:53-57, :80-84), whose last column is this frame's
            [~, offsets, poses] = vohip_mex('frame_bundle_adjust', double(T_W_C));
            for k = 1:numel(offsets)
                T = reshape(poses(:, k), 4, 4);
                S.est_rot(:, end + offsets(k)) = reshape(T(1:3, 1:3), 9, 1);
                S.est_trans(:, end + offsets(k)) = T(1:3, 4);
            end
        end
        function [T_W_C, info] = step(obj, image, next_image)
            if nargin < 3, next_image = []; end
            u8 = [];
            if obj.HaveCandidates, u8 = rand(obj.MapStream, 8, obj.Trials); end
            s = rng;                       % the localisation draws come from the global stream
            have = 104;
            u3 = rand(3, have);
            [T_W_C, info, status] = vohip_mex('frame_step', image, u3, u8, next_image);
            while status ~= 0 && have < 15000     % the adaptive rule wants more hypotheses: same draws, then more
                have = min(max(8 * have, ceil(info.num_iterations) + 8), 15000);
                rng(s);
                u3 = rand(3, have);
                [T_W_C, info, status] = vohip_mex('frame_retry', u3);
            end
            rng(s);
            if info.n_iter > 0, rand(3, info.n_iter); end   % the stream as n_iter executed iterations leave it
            obj.HaveCandidates = info.n_candidates > 0;
        end
        function S = state(~, S)
            [S.P, S.X, S.C, S.F, S.T] = vohip_mex('frame_state');
        end
        function delete(~)
            vohip_mex('window_destroy');
            vohip_mex('frame_destroy');
        end
    end
end

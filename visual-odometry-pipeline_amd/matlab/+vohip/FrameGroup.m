classdef FrameGroup < handle
    % vohip.FrameGroup -- several independent continuous-operation loops (main_demo.m:377-427, is_BA = false)
    % of one image size, camera and parameter set advanced together: ONE library call per frame for all of
    % them, every kernel of the frame launched once with the sequence as a grid dimension.  This is how the
    % independent runs of main_eval.m:20-23 (feature variant x repetition) share a GPU; each sequence's
    % landmarks, candidates and poses are those of its own vohip.FrameLoop, bit for bit.
    %
    %   g = vohip.FrameGroup(K, size(img0), hyper_paras, n_sequences, max_frames);
    %   for k = 1:n_sequences
    %       g.add(k, images_k, S_k.X(1:3, :), S_k.P, T_W_C_k, 'Harris', struct('MinQuality', 2e-4));
    %   end                                  % images_k{1}: the last bootstrap image (main_demo.m:342-344)
    %   while g.active()
    %       [T_W_C, info, took_part] = g.step();     % T_W_C(:, :, k), info(k) of sequence k
    %       ...                                      % main_demo.m:401-403 per sequence that took part
    %   end
    %   S_k = g.state(k, S_k);
    %
    % Random numbers: every sequence has a RandStream of its own for the localisation draws (the
    % reference's global stream serves one run at a time) and one for the fundamental-matrix trials; a
    % localisation stream ends each frame advanced by exactly the iterations the adaptive rule executed.
    properties
        N                    % sequences
        Trials = 500         % update_landmarks.m:13-14
        Streams              % cell: RandStream of the localisation draws per sequence
        MapStreams           % cell: RandStream of the fundamental-matrix trials per sequence
        Frames               % images held per sequence
        Done = 0             % frames processed
        Stopped              % sequence stopped with an error (as the reference's script would)
        HaveCandidates = false
    end
    methods
        function obj = FrameGroup(K, image_size, hyper_paras, n_sequences, max_frames)
            p = struct('r_discard_redundant', hyper_paras.r_discard_redundant, ...
                       'angle_threshold_deg', hyper_paras.angle_threshold, ...
                       'min_depth', hyper_paras.min_depth, 'max_depth', hyper_paras.max_depth, ...
                       'refine_pose', double(hyper_paras.is_refine_pose), 'is_BA', 0);
            vohip_mex('group_create', double(K), image_size(1), image_size(2), p, n_sequences, max_frames);
            obj.N = n_sequences;
            obj.Streams = cell(1, n_sequences);
            obj.MapStreams = cell(1, n_sequences);
            obj.Frames = zeros(1, n_sequences);
            obj.Stopped = false(1, n_sequences);
        end
        function add(obj, k, images, X, P, T_W_C, feature_extract, feature_params, seed)
            % images: cell array of uint8 images, or the index of a sequence added before that holds the same ones
            det = -1; mq = 0; mc = 0.2;
            if nargin > 6 && ~isempty(feature_extract)   % genKeypoints.m:17-19, main_eval.m:165-174
                switch feature_extract
                    case 'Harris',   det = 0; mq = 0.01;
                    case 'MinEigen', det = 1; mq = 0.01;
                    case 'FAST',     det = 2; mq = 0.1;
                    case 'SURF',     det = 3; mq = 0; mc = 1000;   % (the last argument then carries MetricThreshold)
                    case 'BRISK',    det = 4; mq = 0.1;
                    otherwise, error('vohip:FrameGroup', 'detector %s is not built natively', feature_extract);
                end
                if nargin > 7 && isfield(feature_params, 'MinQuality'), mq = feature_params.MinQuality; end
                if nargin > 7 && isfield(feature_params, 'MinContrast'), mc = feature_params.MinContrast; end
                if nargin > 7 && det == 3 && isfield(feature_params, 'MetricThreshold'), mc = feature_params.MetricThreshold; end
            end
            vohip_mex('group_prepare', k, images, det, mq, mc);
            vohip_mex('group_reset', k, double(X), double(P), double(T_W_C));
            if nargin < 9, seed = k; end
            obj.Streams{k} = RandStream('mt19937ar', 'Seed', seed);
            obj.MapStreams{k} = RandStream('mt19937ar', 'Seed', seed + 7919);
            if iscell(images), obj.Frames(k) = numel(images); else, obj.Frames(k) = obj.Frames(images); end
            obj.Stopped(k) = false;
            obj.Done = 0;
            obj.HaveCandidates = false;
        end
        function a = active(obj)
            a = find(~obj.Stopped & obj.Frames > obj.Done + 1);
        end
        function [T_W_C, info, took_part] = step(obj)
            act = obj.active();
            have = 104;
            u3 = zeros(3, have, obj.N);
            u8 = [];
            if obj.HaveCandidates, u8 = zeros(8, obj.Trials, obj.N); end
            states = cell(1, obj.N);
            for k = act
                if obj.HaveCandidates, u8(:, :, k) = rand(obj.MapStreams{k}, 8, obj.Trials); end
                states{k} = obj.Streams{k}.State;
                u3(:, :, k) = rand(obj.Streams{k}, 3, have);
            end
            [T_W_C, info, status] = vohip_mex('group_step', u3, u8);
            obj.Done = obj.Done + 1;
            took_part = false(1, obj.N);
            for k = act
                drawn = have; st = status(k); inf_k = info(k);
                while st == 4 && drawn < 15000    % (4 = VOHIP_SAMPLES_EXHAUSTED) the same draws, then more
                    drawn = min(max(8 * drawn, ceil(inf_k.num_iterations) + 8), 15000);
                    obj.Streams{k}.State = states{k};
                    [T_k, inf_k, st] = vohip_mex('group_retry', k, rand(obj.Streams{k}, 3, drawn));
                    T_W_C(:, :, k) = T_k;
                    info(k) = inf_k;
                end
                obj.Streams{k}.State = states{k};   % the stream as n_iter executed iterations leave it
                if inf_k.n_iter > 0, rand(obj.Streams{k}, 3, inf_k.n_iter); end
                if st ~= 0
                    obj.Stopped(k) = true;           % where the reference stops with an error (main_demo.m:389-391)
                else
                    took_part(k) = true;
                end
            end
            if ~isempty(act), obj.HaveCandidates = info(act(1)).n_candidates > 0; end
        end
        function S = state(~, k, S)
            [S.P, S.X, S.C, S.F, S.T] = vohip_mex('group_state', k);
        end
        function delete(~)
            vohip_mex('group_destroy');
        end
    end
end

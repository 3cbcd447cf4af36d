classdef PointTracker < handle
    % vohip.PointTracker -- GPU stand-in for vision.PointTracker with the protocol the pipeline
    % uses: constructor name-value pairs, initialize(obj, points, I), [points, validity] = obj(I)
    % (also step(obj, I)), release(obj).  Replace the two constructor lines main_demo.m:342-343 and
    % :348-349 by vohip.PointTracker(...) with the same arguments.
    properties
        BlockSize = [31 31]
        NumPyramidLevels = 3
        MaxIterations = 30
        MaxBidirectionalError = inf
    end
    properties (Access = private)
        Points = zeros(0, 2)
        Valid = false(0, 1)
        PrevSlot = -1
    end
    methods
        function obj = PointTracker(varargin)
            for k = 1:2:numel(varargin)
                obj.(varargin{k}) = varargin{k + 1};
            end
        end
        function initialize(obj, points, I)
            obj.PrevSlot = vohip.PointTracker.slotFor(obj, 0);
            vohip_mex('pyramid_build', obj.PrevSlot, I, obj.NumPyramidLevels);
            obj.Points = double(points);
            obj.Valid = true(size(points, 1), 1);
        end
        function [points, validity] = step(obj, I)
            cur = vohip.PointTracker.slotFor(obj, 1);
            vohip_mex('pyramid_build', cur, I, obj.NumPyramidLevels);
            e = obj.MaxBidirectionalError; if isinf(e), e = 1e300; end
            [points, validity] = vohip_mex('klt_track', obj.PrevSlot, cur, obj.Points, obj.BlockSize(1), ...
                obj.MaxIterations, e);
            validity = validity & obj.Valid;   % a lost point stays lost until initialize()
            obj.Valid = validity;
            obj.Points = points;
            vohip.PointTracker.slotFor(obj, -1);   % swap roles: current becomes previous
            obj.PrevSlot = cur;
        end
        function varargout = subsref(obj, s)
            % tracker(I) functor call, as update_landmarks.m:10 uses it
            if strcmp(s(1).type, '()') && isscalar(s)
                [varargout{1:max(nargout, 1)}] = step(obj, s(1).subs{:});
            else
                [varargout{1:nargout}] = builtin('subsref', obj, s);
            end
        end
        function release(obj)
            obj.Points = zeros(0, 2);
        end
    end
    methods (Static, Access = private)
        function slot = slotFor(obj, which)
            % each tracker owns two of the context's eight pyramid slots and alternates between them
            persistent owners flip
            if isempty(owners), owners = {}; flip = []; end
            id = find(cellfun(@(o) o == obj, owners), 1);
            if isempty(id)
                owners{end + 1} = obj; flip(end + 1) = 0; id = numel(owners);
                assert(id <= 4, 'vohip.PointTracker: at most 4 trackers per session');
            end
            if which < 0, flip(id) = 1 - flip(id); slot = -1; return; end
            slot = 2 * (id - 1) + xor(flip(id), which);
        end
    end
end
